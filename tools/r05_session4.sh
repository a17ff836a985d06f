set -x
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "deferred or uniform_timestep or cfg_pair or sd15_unet or tiny" 2>&1 | tail -15
python tools/quick_unet_time.py 2 20 2>&1 | tail -4
python tools/quick_unet_time.py 16 10 2>&1 | tail -4
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/q2 -o q -- python $R/tools/unet_gap.py run 2 > /dev/null 2>&1; python $R/tools/unet_gap.py parse /tmp/q2; python $R/tools/trace_sequence.py /tmp/q2 20 > $R/gpurun_out/seq_b2_new.txt
