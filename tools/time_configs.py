"""Functional + timing run of the BASELINE.json parity-test configurations at full model size on one GPU
(synthetic weights): c3 = SD1.5 runway-inpaint 768x768 batch 4 (VAE encode + 9-channel UNet), c4 = SDXL-base UNet at
1024x1024 (128x128 latents), the per-GPU share of batch 16 over 8 GPUs (2 images), 30 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gyre_amd.pipeline import GyrePipeline
from gyre_amd import schedulers as S

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["c3", "c4"]


def fill(module, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for k, p in module.named_parameters():
            if p.ndim > 1:
                p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
            elif "norm" in k and k.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    module._invalidate()
    return module


def timed(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


vae = fill(GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).to(dev), 1)
if "c3" in which:
    unet = fill(GyreHipUNet(gcfg.sd15_unet(9)).to(torch.bfloat16).to(dev), 0)
    pipe = GyrePipeline(unet, vae, device=dev)
    g = torch.Generator().manual_seed(0)
    text, unc = torch.randn(4, 77, 768, generator=g), torch.randn(4, 77, 768, generator=g)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 768), torch.linspace(0, 1, 768), indexing="ij")
    image = torch.stack([yy, xx, (yy + xx) / 2])[None]
    mask = torch.zeros(1, 1, 768, 768); mask[:, :, 192:576, 192:576] = 1
    dt, img = timed(lambda: pipe(seeds=[1, 2, 3, 4], text_embeddings=text, uncond_embeddings=unc, height=768, width=768,
                                 num_inference_steps=64, sampler="euler_a", image=image.to(dev), mask_image=mask.to(dev),
                                 strength=1.0 - 1e-6), 1)
    print(f"c3 SD1.5 runway-inpaint 768x768 batch 4, 64 steps euler_a (CFG): {dt:.2f} s/batch -> {4/dt:.2f} images/s; "
          f"finite={bool(torch.isfinite(img).all())} evals={pipe.last_unet_evals}")
    enc_t, _ = timed(lambda: vae.encode(image.to(dev).repeat(4, 1, 1, 1) * 2 - 1).latent_dist.mean, 3)
    dec_t, _ = timed(lambda: vae.decode(torch.randn(4, 4, 96, 96, device=dev)).sample, 3)
    print(f"   VAE encode 4x768x768: {enc_t*1e3:.1f} ms ({4*2*1.305/enc_t:.0f} TFLOP/s)   decode: {dec_t*1e3:.1f} ms ({4*2*2.877/dec_t:.0f} TFLOP/s)")
    del unet, pipe
if "c4" in which:
    cfg = gcfg.sdxl_unet()
    unet = fill(GyreHipUNet(cfg).to(torch.bfloat16).to(dev), 2)
    B = 2
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(2 * B, 77, 2048, generator=g).to(dev)
    ac = {"text_embeds": torch.randn(2 * B, 1280, generator=g).to(dev),
          "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * (2 * B), device=dev)}
    x = torch.randn(2 * B, 4, 128, 128, device=dev)
    t = torch.full((2 * B,), 500, device=dev)
    dt, out = timed(lambda: unet(x, t, encoder_hidden_states=ctx, added_cond_kwargs=ac).sample, 3)
    # SDXL-base UNet: ~2.6 B params, analytic cost ~ 6.0 TFLOP per sample-forward at 128x128 latents (approx.)
    print(f"c4 SDXL-base UNet CFG forward, batch {B} (x2 CFG) @128x128 latents: {dt*1e3:.1f} ms; finite={bool(torch.isfinite(out).all())}; "
          f"30-step image batch of {B}: ~{30*dt:.2f} s -> {B/(30*dt):.2f} images/s/GPU (UNet only)")
