#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_ar.py tests/test_gpu_gemm_sm.py -q -x 2>&1 | tail -3 > gpurun_out/s10.log
COLD=1 CONVS=0 BITS=0 timeout 600 python tools/ring_bench.py 2>&1 | grep "res=1" >> gpurun_out/s10.log
timeout 600 python tools/ab_unet.py 0 >> gpurun_out/s10.log 2>&1
B=2 timeout 600 python tools/ab_unet.py 0 >> gpurun_out/s10.log 2>&1
