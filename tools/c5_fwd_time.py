"""Dev tool: UNet forward at batch 8 / 16 with ToMe, plain and activation-keeping (gyre_unet_vjp_begin) - what a batch-16 guided
evaluation would buy config 5."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
from bench import fill_synthetic_on_device

dev = "cuda:0"
unet = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
fill_synthetic_on_device(unet, 1)
unet.set_tome(1024)
g = torch.Generator(device=dev).manual_seed(0)
for B in (8, 16):
    x = torch.randn(B, 4, 64, 64, device=dev, generator=g)
    ctx = torch.randn(B, 77, 768, device=dev, generator=g).to(torch.bfloat16)
    t = torch.full((B,), 500, dtype=torch.int64, device=dev)
    for mode in ("plain", "keep"):
        def run():
            if mode == "plain":
                with torch.no_grad():
                    return unet(x, t, encoder_hidden_states=ctx).sample
            xr = x.detach().requires_grad_()
            with torch.enable_grad():
                return unet(xr, t, encoder_hidden_states=ctx).sample
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run()
        torch.cuda.synchronize()
        print(f"B={B} {mode}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
