cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 300 python tools/r06_xattn.py 2>&1 | grep -v amdgpu.ids
  timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "per_level or full_size or cfg_pair or sd_like" 2>&1 | tail -8
  timeout 600 python tools/ab_unet.py 0x2000 0 2>&1 | grep -v amdgpu.ids
  B=2 timeout 600 python tools/ab_unet.py 0x2000 0 2>&1 | grep -v amdgpu.ids
) > gpurun_out/r06_xattn.txt 2>&1
