import os, sys
sys.path.insert(0, os.getcwd())
import torch
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
dev = "cuda:0"
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).load_synthetic(0).to(dev)
vae = GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).load_synthetic(1).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(16, 4, 64, 64, device=dev, generator=g); t = torch.full((16,), 500, device=dev); ctx = torch.randn(16, 77, 768, device=dev, generator=g)
z = torch.randn(2, 4, 64, 64, device=dev, generator=g)
out = net(x, t, encoder_hidden_states=ctx).sample
img = vae.decode(z).sample
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(4):
    e0.record()
    for _ in range(5): net(x, t, encoder_hidden_states=ctx)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
tag = sys.argv[1]
print(tag, "UNet forward ms:", " ".join(f"{v:.2f}" for v in ts))
path = "/tmp/gn_ab_ref.pt"
if os.path.exists(path):
    ref = torch.load(path)
    print(tag, "bits equal to the other run: unet", bool(torch.equal(ref[0], out.cpu())), "vae", bool(torch.equal(ref[1], img.cpu())))
else:
    torch.save((out.cpu(), img.cpu()), path)
