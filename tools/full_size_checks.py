"""Dev/validation tool (GPU): full-size runs outside the bench configuration.
  * SD1.5 UNet at 96x96 latents (768^2) and SDXL UNet at 128x128 latents (1024^2): planner attention path
    (folded, software-pipelined v3 where it applies) against the plain v2 kernel on the same weights/inputs
  * config 3: runway-inpaint 768^2, batch 4, with and without the hires fix (few steps), finite + timing
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib, config as gcfg
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gyre_amd.pipeline import GyrePipeline

dev = "cuda:0"
L = _lib.lib()


def fill(m, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.ndim > 1:
                p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
            elif k.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    m._invalidate()
    return m


def compare(name, net, x, t, ctx, **kw):
    outs = {}
    for var in (0, 2):
        L.gyre_debug_force_attn_variant(var)
        outs[var] = net(x, t, encoder_hidden_states=ctx, **kw).sample
        torch.cuda.synchronize()
    L.gyre_debug_force_attn_variant(0)
    err = float((outs[0] - outs[2]).norm() / outs[2].norm())
    t0 = time.time()
    for _ in range(3):
        net(x, t, encoder_hidden_states=ctx, **kw)
    torch.cuda.synchronize()
    print(f"{name}: planner vs plain-kernel attention rel-L2 {err:.2e}, finite {bool(torch.isfinite(outs[0]).all())}, "
          f"{(time.time() - t0) / 3 * 1e3:.1f} ms / forward")
    assert err < 3e-2      # random-weight UNet amplifies rounding differences (cf. 1.4e-2 between batch splits)


g = torch.Generator(device=dev).manual_seed(3)
unet = fill(GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev), 0)
compare("SD1.5 96x96 B=4", unet, torch.randn(4, 4, 96, 96, device=dev, generator=g), torch.full((4,), 500, device=dev),
        torch.randn(4, 77, 768, device=dev, generator=g))
compare("SD1.5 72x56 B=2 (ragged)", unet, torch.randn(2, 4, 72, 56, device=dev, generator=g), torch.full((2,), 500, device=dev),
        torch.randn(2, 154, 768, device=dev, generator=g))
del unet
xl = fill(GyreHipUNet(gcfg.sdxl_unet()).to(torch.bfloat16).to(dev), 1)
added = {"text_embeds": torch.randn(2, 1280, device=dev, generator=g), "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * 2, device=dev)}
compare("SDXL 128x128 B=2", xl, torch.randn(2, 4, 128, 128, device=dev, generator=g), torch.full((2,), 500, device=dev),
        torch.randn(2, 77, 2048, device=dev, generator=g), added_cond_kwargs=added)
vae_xl = fill(GyreHipVAE(gcfg.sdxl_vae()).to(torch.bfloat16).to(dev), 5)
pxl = GyrePipeline(xl, vae_xl, None, device=dev)
tx = torch.randn(2, 77, 2048, generator=torch.Generator().manual_seed(1)); un = torch.randn(2, 77, 2048, generator=torch.Generator().manual_seed(2))
for rep in range(2):
    t0 = time.time()
    img = pxl(seeds=[1, 2], text_embeddings=tx, uncond_embeddings=un, height=1024, width=1024, num_inference_steps=30,
              sampler="euler_a", guidance_scale=5.0, added_cond={k: v.cpu() for k, v in added.items()})
    torch.cuda.synchronize()
    print(f"config 4 per-GPU share (SDXL 1024^2, 2 images, 30 steps euler_a, CFG): {time.time() - t0:.2f} s, "
          f"finite {bool(torch.isfinite(img).all())}, shape {tuple(img.shape)}")
del xl, vae_xl, pxl
torch.cuda.empty_cache()

inp = fill(GyreHipUNet(gcfg.sd15_unet(9)).to(torch.bfloat16).to(dev), 2)
vae = fill(GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).to(dev), 3)
pipe = GyrePipeline(inp, vae, None, device=dev)
yy, xx = torch.meshgrid(torch.linspace(0, 1, 768), torch.linspace(0, 1, 768), indexing="ij")
image = torch.stack([xx, yy, (xx + yy) / 2])[None]
mask = torch.zeros(1, 1, 768, 768); mask[:, :, 192:576, 192:576] = 1
text = torch.randn(4, 77, 768, generator=torch.Generator().manual_seed(1)); unc = torch.randn(4, 77, 768, generator=torch.Generator().manual_seed(2))
for hires in (False, True):
    t0 = time.time()
    img = pipe(seeds=[1, 2, 3, 4], text_embeddings=text, uncond_embeddings=unc, height=768, width=768, num_inference_steps=8,
               sampler="dpmpp_2m", image=image, mask_image=mask, strength=1.0, hires_fix=hires)
    torch.cuda.synchronize()
    print(f"config 3 (768^2 runway inpaint, B=4, 8 steps, hires_fix={hires}): {time.time() - t0:.2f} s, evals {pipe.last_unet_evals}, "
          f"finite {bool(torch.isfinite(img).all())}, shape {tuple(img.shape)}")
