cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( echo "== f16 flavour: the five failures + full runs"
  GYRE_STORAGE=f16 timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_vjp.py tests/test_gpu_full_runs.py -q -m gpu -k "nchw or dtypes or controlnet or bf16_io or full_run" 2>&1 | tail -25
  echo "== flavour test (in-process part)"
  timeout 600 python -m pytest tests/test_gpu_f16_flavour.py -q -m gpu -k both_flavours 2>&1 | tail -8
  echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
  cat gpurun_out/full_run_psnr.json
) > gpurun_out/r06_f16_second.txt 2>&1
python bench.py --dtype fp16 --no-cpu-baseline > gpurun_out/r06_bench_fp16.json 2> gpurun_out/r06_bench_fp16.err
python bench.py --no-cpu-baseline > gpurun_out/r06_bench_bf16_samebox.json 2> gpurun_out/r06_bench_bf16.err
