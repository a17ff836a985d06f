#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== gemm tests" > gpurun_out/s2.log
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -k "linear or gemm or tile or qkv or geglu or layernorm or rowstat or colstat" 2>&1 | tail -5 >> gpurun_out/s2.log
echo "== ring bench cold" >> gpurun_out/s2.log
COLD=1 timeout 900 python tools/ring_bench.py >> gpurun_out/s2.log 2>&1
echo "== ring bench cold forced 128x160 / 256x320" >> gpurun_out/s2.log
COLD=1 FORCE=8,4,5 BITS=0x1000000,0,0x2000000 timeout 900 python tools/ring_bench.py >> gpurun_out/s2.log 2>&1
echo "== ab_unet" >> gpurun_out/s2.log
timeout 600 python tools/ab_unet.py 0 0x1000000 0x2000000 >> gpurun_out/s2.log 2>&1
B=2 timeout 600 python tools/ab_unet.py 0 0x1000000 0x2000000 >> gpurun_out/s2.log 2>&1
