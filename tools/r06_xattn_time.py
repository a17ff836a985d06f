import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib, config as gcfg
from gyre_amd.modules import GyreHipUNet
import bench
dev = "cuda:0"; L = _lib.lib()
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev); bench.fill_synthetic_on_device(net, 0)
B = 16
x = torch.randn(B, 4, 64, 64, device=dev); t = torch.full((B,), 500, device=dev); ctx = torch.randn(B, 77, 768, device=dev)
for bits in (0x2000, 0):
    L.gyre_debug_gemm_ablation(bits)
    for _ in range(3): net(x, t, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    _lib.prof_enable(None)
    for _ in range(5): net(x, t, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    c = _lib.prof_collect(); _lib.prof_enable([])
    tot = sum(v["ms"] for v in c.values())
    print(f"bits {bits:#x}: sum of classes {tot / 5:.3f} ms per call")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]["ms"]):
        if k.startswith(("k_xattn", "k_attn", "k_gemm8<256, 320, 4, 2, 0")): print(f"   {k:32s} {v['launches'] / 5:6.1f} launches {v['ms'] / 5 * 1e3:8.1f} us per call  {v['ms'] / v['launches'] * 1e3:7.1f} us each")
L.gyre_debug_gemm_ablation(0)
