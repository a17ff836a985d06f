"""Dev tool: the VAE decode under a rocprofv3 kernel trace (see tools/unet_gap.py / tools/trace_sequence.py):
  cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/v -o v -- python tools/vae_gap.py [B]
  python tools/trace_sequence.py /tmp/v 10
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipVAE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
vae = GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in vae.named_parameters():
        if p.ndim > 1: p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
        elif k.endswith("weight"): p.fill_(1.0)
        else: p.zero_()
vae._invalidate()
z = torch.randn(B, 4, 64, 64, device=dev, generator=g)
for _ in range(2): vae.decode(z).sample
torch.cuda.synchronize()
m = torch.zeros(1, dtype=torch.float64, device=dev)
m.fill_(1.0)
for _ in range(10): vae.decode(z).sample
m.fill_(2.0)
torch.cuda.synchronize()
