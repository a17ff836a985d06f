"""Quick timing of the native SD1.5 UNet forward (dev tool, not the bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg, weights, _lib
from gyre_amd.modules import GyreHipUNet, GyreHipVAE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
t0 = time.time()
cfg = gcfg.sd15_unet()
net = GyreHipUNet(cfg)
# values do not matter for timing: draw on the GPU
with torch.no_grad():
    g = torch.Generator(device=dev).manual_seed(0)
    net = net.to(torch.bfloat16).to(dev)
    for k, p in net.named_parameters():
        fan = p[0].numel() if p.ndim > 1 else 1
        if p.ndim > 1:
            p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / fan ** 0.5)
        elif k.endswith("weight"):
            p.fill_(1.0)
        else:
            p.zero_()
net._invalidate()
print("weights ready", time.time() - t0)
x = torch.randn(B, 4, 64, 64, device=dev, generator=g)       # seeded: the output hash below compares library builds across processes
t = torch.full((B,), 500, device=dev)
ctx = torch.randn(B, 77, 768, device=dev, generator=g)
out = net(x, t, encoder_hidden_states=ctx).sample
torch.cuda.synchronize()
print("first forward ok", time.time() - t0, "finite:", bool(torch.isfinite(out).all()), "launches", _lib.lib().gyre_last_launch_count())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = net(x, t, encoder_hidden_states=ctx).sample
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
import hashlib
print("output sha1", hashlib.sha1(out.float().cpu().numpy().tobytes()).hexdigest()[:16])      # A/B of two library builds: same bits?
print(f"UNet forward B={B}: {ms:.2f} ms  -> {B*0.803/ms:.1f} TFLOP/s effective; ws={net._ws.numel()/2**20:.0f} MiB")
vae = GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).to(dev)
with torch.no_grad():
    for k, p in vae.named_parameters():
        fan = p[0].numel() if p.ndim > 1 else 1
        if p.ndim > 1:
            p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / fan ** 0.5)
        elif k.endswith("weight"):
            p.fill_(1.0)
        else:
            p.zero_()
vae._invalidate()
Bv = max(1, B // 2)
z = torch.randn(Bv, 4, 64, 64, device=dev)
img = vae.decode(z).sample
torch.cuda.synchronize()
e0.record()
for _ in range(3):
    img = vae.decode(z).sample
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"VAE decode B={Bv}: {ms:.2f} ms -> {Bv*2.515/ms:.1f} TFLOP/s effective; finite {bool(torch.isfinite(img).all())}")
