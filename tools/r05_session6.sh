#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== kernel tests (ring / blocked / groupnorm)" > gpurun_out/s6.log
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -k "ring_loop or blocked or groupnorm or gn_" 2>&1 | tail -4 >> gpurun_out/s6.log
timeout 600 python tools/ab_unet.py 0 >> gpurun_out/s6.log 2>&1
B=2 timeout 600 python tools/ab_unet.py 0 >> gpurun_out/s6.log 2>&1
MD=gpurun_out/r05a_unet_layers.md timeout 600 python tools/unet_layers.py > gpurun_out/r05a_unet_layers.txt 2>&1
B=2 MD=gpurun_out/r05a_unet_layers_b2.md timeout 600 python tools/unet_layers.py > gpurun_out/r05a_unet_layers_b2.txt 2>&1
