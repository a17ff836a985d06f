"""Dev tool: per-kernel-class HIP-event time of one SD1.5 UNet forward (B=16, 64x64 latents) under
gyre_debug_gemm_ablation flag values (default 0): `python tools/unet_classes.py 0 0x400`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg, _lib
from gyre_amd.modules import GyreHipUNet

flags = [int(a, 0) for a in sys.argv[1:]] or [0]
B = int(os.environ.get("B", "16")); H = int(os.environ.get("LAT", "64"))
dev = "cuda:0"; L = _lib.lib()
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in net.named_parameters():
        if p.ndim > 1: p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
        elif k.endswith("weight"): p.fill_(1.0)
        else: p.zero_()
net._invalidate()
x = torch.randn(B, 4, H, H, device=dev); t = torch.full((B,), 500, device=dev); ctx = torch.randn(B, 77, 768, device=dev)
for f in flags:
    L.gyre_debug_gemm_ablation(f)
    for _ in range(2): net(x, t, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    _lib.prof_enable(None)
    n = 3
    for _ in range(n): net(x, t, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    prof = _lib.prof_collect(); _lib.prof_enable([])
    tot = sum(v["ms"] for v in prof.values()) / n
    print(f"== flags {f:#x}: sum of timed classes {tot:.2f} ms per forward")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        print(f"  {k:34s} launches {v['launches'] // n:4d}  {v['ms'] / n:7.3f} ms  {v['flops'] / max(v['ms'], 1e-9) / 1e9:7.0f} TFLOP/s  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:7.0f} GB/s")
L.gyre_debug_gemm_ablation(0)
