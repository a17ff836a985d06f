#!/bin/bash
# round 5, GPU session 1: L2/HBM load-path microbenchmark, attention correctness + concurrency probe + timing
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/l2_bw.hip -o /tmp/l2_bw 2>/dev/null && timeout 600 /tmp/l2_bw > gpurun_out/r05_l2_bw.txt 2>&1
echo "== attention tests" > gpurun_out/s1.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -5 >> gpurun_out/s1.log
echo "== redo probe" >> gpurun_out/s1.log
timeout 600 python tools/attn_redo_probe.py >> gpurun_out/s1.log 2>&1
echo "== attention variants (0 = default optimistic, 7 = checked)" >> gpurun_out/s1.log
timeout 300 python tools/attn_variants.py 5 >> gpurun_out/s1.log 2>&1
echo "== thread probe default (optimistic)" >> gpurun_out/s1.log
REPS=90 ATTN=0 timeout 1200 python tools/thread_probe.py 2>&1 | grep -v "mismatches \[\]" | tail -15 >> gpurun_out/s1.log
echo "== ab_unet: default vs checked (variant 7 in bits 24-27)" >> gpurun_out/s1.log
timeout 600 python tools/ab_unet.py 0 0x7000000 >> gpurun_out/s1.log 2>&1
B=2 timeout 600 python tools/ab_unet.py 0 0x7000000 >> gpurun_out/s1.log 2>&1
