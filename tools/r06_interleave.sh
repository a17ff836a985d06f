# round 6: ring requests issued from inside the K step (k_gemm8 128x160) - A/B of two library builds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OLD=gyre_amd/build/libgyre_hip_prev.so; NEW=gyre_amd/libgyre_hip.so
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_sm.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -3
  for L in $OLD $NEW; do
    echo "== ring_bench COLD $(basename $L)"
    GYRE_HIP_LIB=$PWD/$L COLD=1 BITS=0 timeout 600 python tools/ring_bench.py 2>&1 | grep -v amdgpu.ids
  done
  bash tools/ab_libs.sh $OLD $NEW 20 2>&1 | grep -v amdgpu.ids
) > gpurun_out/r06_interleave.txt 2>&1
