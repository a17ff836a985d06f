cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( echo "== bf16 flavour (regression of the HDT edits + new tests)"
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_ar.py tests/test_gpu_gemm_sm.py tests/test_gpu_models.py tests/test_gpu_threads.py -q -m gpu 2>&1 | tail -15
  echo "== f16 flavour"
  GYRE_STORAGE=f16 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_ar.py tests/test_gpu_gemm_sm.py tests/test_gpu_models.py tests/test_gpu_properties.py tests/test_gpu_vjp.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -60
) > gpurun_out/r06_f16_first.txt 2>&1
