"""Round 6: the fused cross-attention block (kernels_xattn.hip) inside the SD1.5 UNet: output vs the three-launch chain (tuning bit 13),
launch counts, and the per-level parity against the fp32 oracle is tests/test_gpu_models.py's job."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib, config as gcfg
from gyre_amd.modules import GyreHipUNet
import bench
dev = "cuda:0"; L = _lib.lib()
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev); bench.fill_synthetic_on_device(net, 0)
for B in (2, 16):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, 4, 64, 64, device=dev, generator=g); t = torch.full((B,), 500, device=dev); ctx = torch.randn(B, 77, 768, device=dev, generator=g)
    outs = {}
    for bits in (0x2000, 0):
        L.gyre_debug_gemm_ablation(bits)
        y = net(x, t, encoder_hidden_states=ctx).sample.float(); torch.cuda.synchronize()
        outs[bits] = (y, L.gyre_last_launch_count())
    L.gyre_debug_gemm_ablation(0)
    a, b = outs[0x2000][0], outs[0][0]
    print(f"B={B}: launches {outs[0x2000][1]} -> {outs[0][1]}; fused vs three-launch rel-L2 {float((a - b).norm() / a.norm()):.3e}; finite {bool(torch.isfinite(b).all())}", flush=True)
