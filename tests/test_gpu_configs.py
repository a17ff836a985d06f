"""BASELINE.json configs 3 and 4 on the HIP path (-m gpu).

config 3  "SD1.5 inpaint/graftedpaint 768x768 (VAE encode + masked latents), batch=4": the 9-channel SD1.5-architecture
          UNet at 96x96 latents (per-level parity vs the fp32 oracle at batch 1, size-independent properties at the full
          batch), the runway-inpaint and grafted-inpaint pipelines (reference unified_pipeline.py:648-696, 2071-2100,
          unet/graft.py:16-56) against the same host flow on the oracle models, and one full-size 768x768 run.
config 4  "SDXL-base 1024x1024": the REAL SDXL-base topology (3 levels 320/640/1280, transformer depth 0/2/10, head dim 64,
          context 2048, text_time conditioning - not in the reference, an extension on the same kernel set): per-level
          parity at 32x32 latents, properties at 128x128.
Tolerances as in tests/test_gpu_models.py (bf16 storage / fp32 accumulate vs fp32 oracle): 2e-2 per level, 3e-2 eps,
image PSNR >= 30 dB."""
import ctypes as C

import pytest
import torch

from gyre_amd import _lib, config as gcfg, weights
from gyre_amd.modules import GyreHipUNet, GyreHipVAE, set_batch_invariant
from gyre_amd.pipeline import GyrePipeline
from gpu_util import HDT, DEV, randn, report
from oracle import models_ref as M
from oracle import pipeline_ref as PR

pytestmark = pytest.mark.gpu


def fill(module, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    with torch.no_grad():
        for k, p in module.named_parameters():
            if p.ndim > 1:
                p.copy_(torch.randn(p.shape, device=DEV, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
            elif "norm" in k and k.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    module._invalidate()
    return module


def taps_parity(cfg, sd, net, x, t, ctx, label, **kw):
    taps = {}
    ref = M.unet_forward(sd, cfg, x, t, ctx, taps=taps, **({"added_cond": kw["ac"]} if "ac" in kw else {}))
    fkw = {"added_cond_kwargs": kw["ac"]} if "ac" in kw else {}
    net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), **fkw)          # uploads weights, creates the handle
    n = len(cfg.block_out_channels)
    names = [f"down{i}" for i in range(n)] + ["mid"] + [f"up{i}" for i in range(n)]
    bufs = {k: torch.empty(taps[k].shape, dtype=torch.float32, device=DEV) for k in names}
    L = _lib.lib()
    for k, b in bufs.items():
        _lib.check(L.gyre_unet_debug_tap(C.c_void_p(net._handle), k.encode(), C.c_void_p(b.data_ptr()), b.numel() * 4))
    out = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), **fkw).sample
    torch.cuda.synchronize()
    for k in names:
        report(f"{label} level {k} {tuple(taps[k].shape)}", bufs[k].cpu(), taps[k], 2e-2)
    report(f"{label} eps", out.cpu(), ref, 3e-2)


# ---- config 3 ------------------------------------------------------------------------------------------------------------
def test_config3_inpaint_unet_per_level_parity():
    """SD1.5-architecture runway-inpaint UNet (9 input channels), 96x96 latents (768x768 px), one sample."""
    cfg = gcfg.sd15_unet(in_channels=9)
    sd = weights.synthetic_state_dict(weights.unet_param_shapes(cfg), 3)
    net = GyreHipUNet(cfg)
    net.load_state_dict(sd)
    net = net.to(DEV)
    x, t, ctx = randn(1, 9, 96, 96, seed=31), torch.tensor([640]), randn(1, 77, 768, seed=32)
    taps_parity(cfg, sd, net, x, t, ctx, "config3 9-ch unet 96x96")


def test_config3_inpaint_unet_properties_full_size():
    """Config-3 UNet call: 4 images x CFG = batch 8, 9 channels, 96x96 latents.  Determinism, sample-permutation
    equivariance (bit-exact) and, under batch-invariant planning, bit-exact batch splits."""
    net = fill(GyreHipUNet(gcfg.sd15_unet(in_channels=9)).to(HDT).to(DEV), 5)
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(8, 9, 96, 96, device=DEV, generator=g)
    ctx = torch.randn(8, 77, 768, device=DEV, generator=g)
    t = torch.full((8,), 555, device=DEV)
    full = net(x, t, encoder_hidden_states=ctx).sample
    assert full.shape == (8, 4, 96, 96) and bool(torch.isfinite(full).all())
    assert torch.equal(full, net(x, t, encoder_hidden_states=ctx.clone()).sample)
    perm = torch.randperm(8, device=DEV, generator=g)
    assert torch.equal(net(x[perm].contiguous(), t, encoder_hidden_states=ctx[perm].contiguous()).sample, full[perm])
    prev = set_batch_invariant(8)
    try:
        inv = net(x, t, encoder_hidden_states=ctx).sample
        for lo, hi in ((0, 4), (4, 8), (2, 3), (7, 8)):
            part = net(x[lo:hi].contiguous(), t[lo:hi], encoder_hidden_states=ctx[lo:hi].contiguous()).sample
            assert torch.equal(part, inv[lo:hi]), (lo, hi)
    finally:
        set_batch_invariant(prev)


@pytest.fixture(scope="module")
def tiny_pair():
    """tiny 4-channel UNet + tiny 9-channel inpaint UNet + tiny VAE, native and oracle-adapter pipelines"""
    from test_host_pipeline import OracleUNet, OracleVAE
    u4, u9, vc = gcfg.tiny_unet(), gcfg.tiny_unet(in_channels=9), gcfg.tiny_vae()
    sd4 = weights.synthetic_state_dict(weights.unet_param_shapes(u4), 0)
    sd9 = weights.synthetic_state_dict(weights.unet_param_shapes(u9), 1)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vc), 2)

    def native(cfg, sd, cls=GyreHipUNet):
        m = cls(cfg)
        m.load_state_dict(sd)
        return m.to(DEV)

    g = torch.Generator().manual_seed(5)
    text = torch.randn(2, 77, u4.cross_attention_dim, generator=g)
    unc = torch.randn(1, 77, u4.cross_attention_dim, generator=g).expand(2, -1, -1).contiguous()
    image = torch.rand(1, 3, 128, 128, generator=g)
    mask = torch.zeros(1, 1, 128, 128)
    mask[:, :, 24:104, 40:112] = 1.0                                           # 1 = repaint (0K1D)
    mk = lambda dev, U, V, **kw: GyrePipeline(U(u4, sd4), V(vc, vsd), device=dev, inpaint_unet=U(u9, sd9), **kw)
    nat = lambda **kw: mk(DEV, native, lambda c, s: native(c, s, GyreHipVAE), **kw)
    ora = lambda **kw: mk("cpu", lambda c, s: OracleUNet(s, c), lambda c, s: OracleVAE(s, c), **kw)
    return nat, ora, text, unc, image, mask


def test_tiny_runway_inpaint_pipeline_psnr(tiny_pair):
    """EnhancedRunwayInpaintMode end to end on the native models (VAE encode of the masked original, 9-channel
    assembly, no per-step blend) vs the same host flow on the fp32 oracle models."""
    nat, ora, text, unc, image, mask = tiny_pair
    for sampler, strength in (("euler_a", 0.8), ("dpmpp_2m", 1.0), ("ddim", 0.6)):
        kw = dict(seeds=[11, 12], text_embeddings=text, uncond_embeddings=unc, height=128, width=128,
                  num_inference_steps=6, sampler=sampler, strength=strength)
        got = nat()(image=image.to(DEV), mask_image=mask.to(DEV), **kw).cpu()
        ref = ora()(image=image, mask_image=mask, **kw)
        p = PR.psnr(got, ref)
        print(f"[parity] tiny runway inpaint {sampler} strength {strength}: PSNR {p:.1f} dB")
        assert got.shape == (2, 3, 128, 128) and p >= 30.0


def test_tiny_grafted_inpaint_pipeline_psnr(tiny_pair):
    """Grafted inpaint ("graftedpaint", reference tests/graftedpaint.py): inpaint UNet -> base UNet hand-over by
    GraftUnets while u crosses [0.1, 0.3] (both UNets evaluated there), Euler-a with churn and a Karras schedule as in
    the reference script."""
    nat, ora, text, unc, image, mask = tiny_pair
    kw = dict(seeds=[21, 22], text_embeddings=text, uncond_embeddings=unc, height=128, width=128, num_inference_steps=12,
              sampler="euler_a", strength=1.0, karras_rho=7.0, churn=0.4)
    pn, po = nat(grafted_inpaint=True), ora(grafted_inpaint=True)
    got = pn(image=image.to(DEV), mask_image=mask.to(DEV), **kw).cpu()
    ref = po(image=image, mask_image=mask, **kw)
    p = PR.psnr(got, ref)
    print(f"[parity] tiny grafted inpaint: PSNR {p:.1f} dB, UNet evals {pn.last_unet_evals}")
    assert pn.last_unet_evals == po.last_unet_evals and 12 < pn.last_unet_evals < 24     # both UNets only inside the blend window
    assert p >= 30.0
    # a custom blend window is honoured: hand-over finished before the first step -> only the base UNet runs
    pn2 = nat(grafted_inpaint={"start": -2.0, "end": -1.0})
    pn2(image=image.to(DEV), mask_image=mask.to(DEV), **kw)
    assert pn2.last_unet_evals == 12


def test_config3_full_size_grafted_inpaint_768():
    """Config 3 at its real size: 768x768, 4 images, SD1.5-architecture 9-channel inpaint UNet grafted onto the 4-channel
    UNet, VAE encode of the (masked) init image, hires fix active (768 > 529 px: natural 64x64 + full 96x96 leaves).
    No oracle at this size: shape, finiteness, bit-determinism and batch independence (reference
    tests/batch_independance.py:15-27) of the whole pipeline."""
    unet = fill(GyreHipUNet(gcfg.sd15_unet()).to(HDT).to(DEV), 0)
    inp = fill(GyreHipUNet(gcfg.sd15_unet(in_channels=9)).to(HDT).to(DEV), 1)
    vae = fill(GyreHipVAE(gcfg.sd15_vae()).to(HDT).to(DEV), 2)
    pipe = GyrePipeline(unet, vae, device=DEV, inpaint_unet=inp, grafted_inpaint=True)
    g = torch.Generator().manual_seed(3)
    text, unc = torch.randn(4, 77, 768, generator=g) * 0.3, torch.randn(1, 77, 768, generator=g).expand(4, -1, -1) * 0.3
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 768), torch.linspace(0, 1, 768), indexing="ij")
    image = torch.stack([yy, xx, (yy + xx) / 2])[None]                          # deterministic gradient image
    mask = torch.zeros(1, 1, 768, 768)
    mask[:, :, 192:576, 192:576] = 1.0                                          # centre square is repainted
    kw = dict(height=768, width=768, num_inference_steps=6, sampler="euler_a", strength=1.0, churn=0.4, karras_rho=7.0,
              image=image.to(DEV), mask_image=mask.to(DEV), output_type="latent")
    seeds = [420420420, 420420421, 420420422, 420420423]
    prev = set_batch_invariant(8)
    try:
        lat = pipe(seeds=seeds, text_embeddings=text, uncond_embeddings=unc.contiguous(), **kw)
        evals = pipe.last_unet_evals
        assert lat.shape == (4, 4, 96, 96) and bool(torch.isfinite(lat).all())
        again = pipe(seeds=seeds, text_embeddings=text, uncond_embeddings=unc.contiguous(), **kw)
        assert torch.equal(lat, again)
        one = pipe(seeds=seeds[2:3], text_embeddings=text[2:3], uncond_embeddings=unc[2:3].contiguous(), **kw)
        assert torch.equal(one, lat[2:3]), "image 2 changed when generated alone"
    finally:
        set_batch_invariant(prev)
    img = pipe.vae_decode(lat)
    assert img.shape == (4, 3, 768, 768) and bool(torch.isfinite(img).all())
    print(f"[config3] 768x768 grafted inpaint, 6 steps: {evals} UNet evaluations (2 UNets x 2 resolutions while blending)")
    assert evals > 6


# ---- config 4 ------------------------------------------------------------------------------------------------------------
def test_config4_sdxl_base_per_level_parity():
    """Real SDXL-base UNet (2.57 B parameters), 32x32 latents, context 77 x 2048, text_time conditioning."""
    cfg = gcfg.sdxl_unet()
    sd = weights.synthetic_state_dict(weights.unet_param_shapes(cfg), 4)
    net = GyreHipUNet(cfg)
    net.load_state_dict(sd)
    net = net.to(HDT).to(DEV)
    sd = {k: v.to(HDT).float() for k, v in sd.items()}              # the oracle sees the same (rounded) weights
    x, t, ctx = randn(1, 4, 32, 32, seed=41), torch.tensor([500]), randn(1, 77, 2048, seed=42) * 0.5
    ac = {"text_embeds": randn(1, 1280, seed=43), "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]])}
    taps_parity(cfg, sd, net, x, t, ctx, "config4 SDXL-base unet 32x32", ac=ac)


def test_config4_sdxl_base_properties_full_size():
    """SDXL-base at 128x128 latents (1024x1024 px), CFG pair of one image = batch 2 (the per-GPU call of config 4:
    16 images over 8 GPUs): finite, bit-deterministic, permutation-equivariant, and the VAE decodes 128x128 latents."""
    cfg = gcfg.sdxl_unet()
    net = fill(GyreHipUNet(cfg).to(HDT).to(DEV), 7)
    g = torch.Generator(device=DEV).manual_seed(8)
    x = torch.randn(4, 4, 128, 128, device=DEV, generator=g)
    ctx = torch.randn(4, 77, 2048, device=DEV, generator=g) * 0.5
    ac = {"text_embeds": torch.randn(4, 1280, device=DEV, generator=g),
          "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * 4, device=DEV)}
    t = torch.full((4,), 700, device=DEV)
    out = net(x, t, encoder_hidden_states=ctx, added_cond_kwargs=ac).sample
    assert out.shape == (4, 4, 128, 128) and bool(torch.isfinite(out).all())
    assert torch.equal(out, net(x, t, encoder_hidden_states=ctx.clone(), added_cond_kwargs=ac).sample)
    perm = torch.tensor([2, 0, 3, 1], device=DEV)
    acp = {k: v[perm].contiguous() for k, v in ac.items()}
    assert torch.equal(net(x[perm].contiguous(), t, encoder_hidden_states=ctx[perm].contiguous(), added_cond_kwargs=acp).sample, out[perm])
    vae = fill(GyreHipVAE(gcfg.sdxl_vae()).to(HDT).to(DEV), 9)
    img = vae.decode(x[:1] / 0.13025).sample
    assert img.shape == (1, 3, 1024, 1024) and bool(torch.isfinite(img).all())


# ------------------------------------------------------------------------------------------------------------------
# BASELINE config 5: SD1.5 512x512 with ToMe token-merged attention + CLIP guidance, batch 8
# ------------------------------------------------------------------------------------------------------------------
def test_config5_sd15_tome_clip_guidance_batch8_full_size():
    """The real SD1.5 topology at 512x512, batch 8, ToMe r = 1024 and CLIP guidance together (reference
    tests/engines.clip.yaml + option `tome`): every guided step runs the conditional stem's native reverse sweep THROUGH the
    merged attention.  No CPU oracle at this size (its backward alone takes minutes): size-independent properties -
    finite, bit-reproducible, guidance and merging both act, evaluation count, and the result of image i does not change
    when the images after it are dropped from the batch EXCEPT through the batch-wide loss history (flat-loss test off)."""
    from types import SimpleNamespace
    from transformers import CLIPConfig, CLIPModel
    from gyre_amd.clipguided import patch_embedding_as_matmul
    ucfg, vcfg = gcfg.sd15_unet(), gcfg.sd15_vae()
    unet = GyreHipUNet(ucfg).load_synthetic(0).to(DEV)
    vae = GyreHipVAE(vcfg).load_synthetic(1).to(DEV)
    torch.manual_seed(0)
    clip = patch_embedding_as_matmul(CLIPModel(CLIPConfig(projection_dim=512)).eval().to(DEV))
    for p_ in clip.parameters():
        p_.requires_grad_(False)
    fe = SimpleNamespace(image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711],
                         size={"shortest_edge": 224})
    pipe = GyrePipeline(unet, vae, device=DEV, clip_model=clip, feature_extractor=fe)
    g = torch.Generator().manual_seed(5)
    B = 8
    text, unc = torch.randn(B, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g).expand(B, -1, -1).contiguous()
    ids = torch.randint(3, 40000, (B, 77), generator=g)
    kw = dict(seeds=list(range(100, 100 + B)), text_embeddings=text, uncond_embeddings=unc, height=512, width=512,
              num_inference_steps=3, guidance_scale=7.5, sampler="dpmpp_2m", output_type="latent")
    ckw = dict(clip_guidance_scale=0.2, clip_input_ids=ids, clip_gradient_threshold=0.0)
    plain = pipe(**kw)
    unet.set_tome(1024)
    merged = pipe(**kw)
    import time
    torch.cuda.synchronize(); t0 = time.time()
    a = pipe(**ckw, **kw)
    torch.cuda.synchronize(); dt = time.time() - t0
    evals, mode = pipe.last_unet_evals, pipe.last_clip_modes[0]
    b = pipe(**ckw, **kw)
    vae_only = pipe(vae_cutouts=4, approx_cutouts=0, **ckw, **kw)
    unet.set_tome(0)
    guided_no_tome = pipe(**ckw, **kw)
    print(f"[config5] B=8 512x512 ToMe 1024 + CLIP guidance: {dt / 3 * 1e3:.0f} ms per guided step (batch 8), loss {mode.lossavg}, "
          f"evals {evals}; |tome - plain| {float((merged - plain).abs().max()):.3f}, |guided - tome| {float((a - merged).abs().max()):.3f}, "
          f"|guided(tome) - guided(no tome)| {float((a - guided_no_tome).abs().max()):.3f}")
    assert a.shape == (B, 4, 64, 64) and bool(torch.isfinite(a).all()) and bool(torch.isfinite(vae_only).all())
    assert torch.equal(a, b)
    assert evals == 2 * 4 and mode.grad_evals == 4            # dpmpp_2m: 3 steps + 1 warm-up evaluation, each guided: g-stem + (u)
    assert float((merged - plain).abs().max()) > 1e-3 and float((a - merged).abs().max()) > 1e-3
    assert float((a - guided_no_tome).abs().max()) > 1e-3 and float((a - vae_only).abs().max()) > 1e-3
