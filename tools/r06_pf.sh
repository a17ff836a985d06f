# round 6: L2 prefetch in the 128-row k_gemm8 ring (tuning bit 13 = off): bit-identity tests, per-shape A/B cold + warm, UNet A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_sm.py -x -q -m gpu 2>&1 | tail -5
  echo "== ring_bench COLD (bits: 0x2000 = no prefetch | 0 = prefetch)"
  COLD=1 CONVS=0 BITS=0x2000,0 timeout 600 python tools/ring_bench.py
  echo "== ring_bench WARM"
  COLD=0 CONVS=0 BITS=0x2000,0 timeout 600 python tools/ring_bench.py
  echo "== UNet A/B batch 16"
  timeout 600 python tools/ab_unet.py 0x2000 0
  echo "== UNet A/B batch 2"
  B=2 timeout 600 python tools/ab_unet.py 0x2000 0
) > gpurun_out/r06_pf.txt 2>&1
