"""Dev tool: UNet forward time with / without batch-invariant planning at several batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet, set_batch_invariant

dev = "cuda:0"
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in net.named_parameters():
        if p.ndim > 1:
            p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
        elif k.endswith("weight"):
            p.fill_(1.0)
        else:
            p.zero_()
net._invalidate()


def t_fwd(B, n=10):
    x = torch.randn(B, 4, 64, 64, device=dev); t = torch.full((B,), 500, device=dev); ctx = torch.randn(B, 77, 768, device=dev)
    net(x, t, encoder_hidden_states=ctx); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        net(x, t, encoder_hidden_states=ctx)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in (2, 4, 8, 16, 32):
    set_batch_invariant(0); a = t_fwd(B)
    set_batch_invariant(16); b = t_fwd(B)
    set_batch_invariant(0)
    print(f"B={B:3d}: default {a:7.3f} ms   invariant(16) {b:7.3f} ms   ({(b/a-1)*100:+.1f}%)")
