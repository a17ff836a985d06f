"""Dev tool: is the UNet permutation-equivariant over the batch (bit-exact)?  python tools/perm_check.py [debug bits ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg, _lib
from gyre_amd.modules import GyreHipUNet
dev = "cuda:0"; L = _lib.lib()
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in net.named_parameters():
        if p.ndim > 1: p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
        elif k.endswith("weight"): p.fill_(1.0)
        else: p.zero_()
net._invalidate()
B = int(os.environ.get("B", "16")); H = int(os.environ.get("LAT", "64"))
x = torch.randn(B, 4, H, H, device=dev, generator=g); ctx = torch.randn(B, 77, 768, device=dev, generator=g)
t = torch.full((B,), 801, device=dev)
perm = torch.randperm(B, device=dev, generator=g)
for bits in [int(a, 0) for a in sys.argv[1:]] or [0]:
    L.gyre_debug_gemm_ablation(bits)
    import ctypes as C
    shapes = {"down0": (320, H // 2), "down1": (640, H // 4), "down2": (1280, H // 8), "down3": (1280, H // 8), "mid": (1280, H // 8),
              "up0": (1280, H // 4), "up1": (1280, H // 2), "up2": (640, H), "up3": (320, H)}
    def run(xx, cc):
        bufs = {}
        net(xx, t, encoder_hidden_states=cc)            # creates the handle / context
        for name, (ch, hw) in shapes.items():
            bufs[name] = torch.zeros(B, ch, hw, hw, device=dev)
            _lib.check(L.gyre_unet_debug_tap(C.c_void_p(net._handle), name.encode(), C.c_void_p(bufs[name].data_ptr()), bufs[name].numel() * 4))
        out = net(xx, t, encoder_hidden_states=cc).sample
        torch.cuda.synchronize()
        return out, bufs
    full, tf = run(x, ctx)
    pf, tp_ = run(x[perm].contiguous(), ctx[perm].contiguous())
    for name in shapes:
        print(f"   tap {name}: permutes exactly: {bool(torch.equal(tp_[name], tf[name][perm]))}")
    d = (pf - full[perm]).float()
    per = d.flatten(1).norm(dim=1) / full[perm].float().flatten(1).norm(dim=1)
    print(f"bits {bits:#x}: equal {bool(torch.equal(pf, full[perm]))}, per-sample rel diff {[round(float(v), 5) for v in per]}  perm {perm.tolist()}")
L.gyre_debug_gemm_ablation(0)
