#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== blocked weights (bit 26 = row-major) cold" > gpurun_out/s4.log
COLD=1 BITS=0x4000000,0 timeout 900 python tools/ring_bench.py >> gpurun_out/s4.log 2>&1
echo "== forced configs, linear only" >> gpurun_out/s4.log
COLD=1 CONVS=0 FORCE=8,4,24,32 BITS=0x4000000,0 timeout 900 python tools/ring_bench.py 2>&1 | grep "K= 1280\|K= 2560\|K= 5120" >> gpurun_out/s4.log
echo "== ab_unet" >> gpurun_out/s4.log
timeout 600 python tools/ab_unet.py 0 0x4000000 >> gpurun_out/s4.log 2>&1
B=2 timeout 600 python tools/ab_unet.py 0 0x4000000 >> gpurun_out/s4.log 2>&1
