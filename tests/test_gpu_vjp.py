"""Input-gradient (VJP) parity (-m gpu): the native backward kernels and the UNet / VAE-decoder reverse sweeps
(gyre_unet_vjp / gyre_vae_decode_vjp, include/gyre_hip.h) against torch autograd over fp32 references - ATen ops for the
kernels, the fp32 oracle (oracle/models_ref.py) for the model graphs.

This is the arithmetic the reference's CLIP-guided mode asks autograd for (gyre/pipeline/unet/clipguided.py:301-338,420:
gradient of a scalar loss with respect to the latents through unet(latents, t) and vae.decode).
Tolerances: bf16 activations / gradients with fp32 accumulation vs fp32: kernels 2e-2 rel-L2, model sweeps 5e-2.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from gyre_amd import _lib
from gyre_amd import config as gcfg
from gyre_amd import weights
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gpu_util import HDT, DEV, bf16_round, randn, release_kept, report, st, vp
from oracle import models_ref as M

pytestmark = pytest.mark.gpu


def dev_bf16(t):
    return t.to(HDT).contiguous().to(DEV)


@pytest.mark.parametrize("B,H,W,C,C1,G,silu,add", [(2, 8, 8, 64, 64, 32, 1, 0), (2, 8, 8, 64, 32, 32, 1, 0),
                                                   (2, 32, 32, 320, 320, 32, 1, 1), (1, 16, 16, 960, 640, 32, 1, 0),
                                                   (2, 16, 16, 128, 128, 32, 0, 1), (3, 5, 7, 64, 64, 32, 1, 0),
                                                   (1, 96, 96, 128, 128, 32, 1, 0)])
def test_groupnorm_bwd(B, H, W, C, C1, G, silu, add):
    L = _lib.lib()
    HW = H * W
    x = bf16_round(randn(B, HW, C, seed=1) * 1.5 + 0.3)
    dy = bf16_round(randn(B, HW, C, seed=2))
    gamma, beta = randn(C, seed=3) * 0.5 + 1.0, randn(C, seed=4) * 0.2
    addend = bf16_round(randn(B, HW, C1, seed=5)) if add else None
    xr = x.clone().requires_grad_()
    y = F.group_norm(xr.permute(0, 2, 1).reshape(B, C, H, W), G, gamma, beta, 1e-5)
    if silu:
        y = F.silu(y)
    (ref,) = torch.autograd.grad(y, xr, dy.permute(0, 2, 1).reshape(B, C, H, W))
    if add:
        ref = torch.cat([ref[..., :C1] + addend, ref[..., C1:]], -1)
    x1, x2 = dev_bf16(x[..., :C1]), (dev_bf16(x[..., C1:]) if C1 < C else None)
    dx = torch.empty(B, HW, C1, dtype=HDT, device=DEV)
    dx2 = torch.empty(B, HW, C - C1, dtype=HDT, device=DEV) if C1 < C else None
    need = L.gyre_op_groupnorm_bwd_workspace(B, HW, C, G)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    _lib.check(L.gyre_op_groupnorm_bwd(st(), vp(x1), vp(x2), C1, B, HW, C, G, vp(gamma.to(DEV)), vp(beta.to(DEV)), 1e-5, silu,
                                       vp(dev_bf16(dy)), vp(dev_bf16(addend)) if add else None, vp(ws), need, vp(dx), vp(dx2)))
    release_kept()
    got = dx.float().cpu() if C1 == C else torch.cat([dx.float().cpu(), dx2.float().cpu()], -1)
    report(f"gn_bwd {B}x{H}x{W}x{C} C1={C1} silu={silu} add={add}", got, ref, 2e-2)


@pytest.mark.parametrize("M_,C,add", [(64, 320, 1), (33, 640, 0), (128, 1280, 1), (7, 64, 0), (16, 2048, 0)])
def test_layernorm_bwd(M_, C, add):
    L = _lib.lib()
    x = bf16_round(randn(M_, C, seed=1) * 2 + 0.5)
    dy = bf16_round(randn(M_, C, seed=2))
    gamma, beta = randn(C, seed=3) * 0.5 + 1.0, randn(C, seed=4) * 0.1
    addend = bf16_round(randn(M_, C, seed=5)) if add else None
    xr = x.clone().requires_grad_()
    (ref,) = torch.autograd.grad(F.layer_norm(xr, (C,), gamma, beta, 1e-5), xr, dy)
    if add:
        ref = ref + addend
    dx = torch.empty(M_, C, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_layernorm_bwd(st(), vp(dev_bf16(x)), vp(dev_bf16(dy)), M_, C, vp(gamma.to(DEV)), 1e-5,
                                       vp(dev_bf16(addend)) if add else None, vp(dx)))
    release_kept()
    report(f"ln_bwd {M_}x{C} add={add}", dx.float().cpu(), ref, 2e-2)


@pytest.mark.parametrize("M_,F_", [(64, 1280), (33, 2560), (5, 32)])
def test_geglu_bwd(M_, F_):
    L = _lib.lib()
    val, gate = bf16_round(randn(M_, F_, seed=1)), bf16_round(randn(M_, F_, seed=2) * 1.5)
    dy = bf16_round(randn(M_, F_, seed=3))
    vr, gr = val.clone().requires_grad_(), gate.clone().requires_grad_()
    dval, dgate = torch.autograd.grad(vr * F.gelu(gr), (vr, gr), dy)

    def pack(v, g):   # packed column order of the GEGLU weight rows: 16 values then their 16 gates (kernels_elem.hip repack)
        return torch.stack([v.reshape(M_, F_ // 16, 16), g.reshape(M_, F_ // 16, 16)], 2).reshape(M_, 2 * F_)
    dpre = torch.empty(M_, 2 * F_, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_geglu_bwd(st(), vp(dev_bf16(pack(val, gate))), vp(dev_bf16(dy)), M_, F_, vp(dpre)))
    release_kept()
    report(f"geglu_bwd {M_}x{F_}", dpre.float().cpu(), pack(dval, dgate), 2e-2)


@pytest.mark.parametrize("B,heads,Nq,Nk,D,cross,presc", [
    (2, 2, 64, 64, 40, 0, 1), (1, 8, 100, 100, 40, 0, 1), (2, 4, 96, 77, 80, 1, 1), (1, 2, 256, 256, 160, 0, 1),
    (2, 3, 130, 130, 64, 0, 0), (1, 1, 144, 144, 512, 0, 0), (1, 5, 1024, 1024, 64, 0, 1), (2, 8, 1024, 77, 40, 1, 1),
    (1, 1, 33, 33, 8, 0, 0),
    # the LDS-DMA kernels of D = 40 (round 6): ragged query and key tiles, keys != queries (ToMe), not prescaled, many key tiles
    (1, 2, 1000, 840, 40, 0, 0), (1, 1, 200, 4100, 40, 0, 1), (2, 2, 65, 127, 40, 0, 1), (1, 3, 128, 64, 40, 0, 1),
    (2, 2, 200, 160, 80, 0, 1), (1, 2, 96, 96, 96, 0, 0)])      # D = 80 (five k-steps) and D = 96 (six) self-attention
def test_attention_bwd(B, heads, Nq, Nk, D, cross, presc):
    """dq / dk / dv of softmax(q k^T / sqrt(D)) v.  With k_prescaled the kernel is handed k' = k * log2(e)/sqrt(D) (what
    the UNet's to_k weights produce) and returns the gradient with respect to k'."""
    _attention_bwd_case(B, heads, Nq, Nk, D, cross, presc)


def test_attention_bwd_large_logits_recentre_the_single_pass_dq_kernel():
    """Scores tens of log2 units apart and growing along the key axis: the one-pass dQ kernel of D = 40 re-centres its reference
    several times per row (kernels_bwd.hip k_attn_bwd_dq_dma, RECENTRE = 20) - same gradients as autograd."""
    _attention_bwd_case(1, 2, 192, 640, 40, 0, 1, qscale=6.0, ramp=True)


def _attention_bwd_case(B, heads, Nq, Nk, D, cross, presc, qscale=1.0, ramp=False):
    L = _lib.lib()
    C_ = heads * D
    c = math.log2(math.e) / math.sqrt(D)
    q = bf16_round(randn(B, Nq, C_, seed=1) * qscale)
    kk = randn(B, Nk, C_, seed=2)
    if ramp:                                                              # later keys score higher: the running reference keeps moving
        kk = kk * torch.linspace(0.2, 2.0, Nk).view(1, Nk, 1)
    k_in = bf16_round(kk * (c if presc else 1.0))     # what the kernel sees
    v = bf16_round(randn(B, Nk, C_, seed=3))
    d_o = bf16_round(randn(B, Nq, C_, seed=4))
    qr, kr, vr = q.clone().requires_grad_(), k_in.clone().requires_grad_(), v.clone().requires_grad_()

    def split(t, n):
        return t.reshape(B, n, heads, D).permute(0, 2, 1, 3)
    logits = split(qr, Nq) @ split(kr, Nk).transpose(-1, -2) * (math.log(2.0) if presc else 1.0 / math.sqrt(D))
    o = (logits.softmax(-1) @ split(vr, Nk)).permute(0, 2, 1, 3).reshape(B, Nq, C_)
    rq, rk, rv = torch.autograd.grad(o, (qr, kr, vr), d_o)
    o_dev = dev_bf16(o.detach())
    dq = torch.zeros(B, Nq, C_, dtype=HDT, device=DEV)
    dk = torch.zeros(B, Nk, C_, dtype=HDT, device=DEV) if not cross else None
    dv = torch.zeros(B, Nk, C_, dtype=HDT, device=DEV) if not cross else None
    need = L.gyre_op_attention_bwd_workspace(B, heads, Nq, Nk, D)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    _lib.check(L.gyre_op_attention_bwd(st(), vp(dev_bf16(q)), C_, vp(dev_bf16(k_in)), C_, vp(dev_bf16(v)), C_, vp(o_dev), C_,
                                       vp(dev_bf16(d_o)), C_, B, heads, Nq, Nk, D, presc, vp(ws), need, vp(dq), C_,
                                       vp(dk), C_, vp(dv), C_))
    release_kept()
    tag = f"attn_bwd B{B} h{heads} {Nq}x{Nk} D{D} presc{presc}" + (" large logits" if ramp else "")
    report(tag + " dq", dq.float().cpu(), rq, 2e-2)
    if not cross:
        report(tag + " dk", dk.float().cpu(), rk, 2e-2)
        report(tag + " dv", dv.float().cpu(), rv, 2e-2)


def _unet(cfg, seed=0):
    sd = weights.synthetic_state_dict(weights.unet_param_shapes(cfg), seed)
    net = GyreHipUNet(cfg)
    net.load_state_dict(sd)
    return net.to(DEV), sd


@pytest.mark.parametrize("inch,H,W,S", [(4, 16, 16, 77), (9, 16, 24, 77), (4, 9, 15, 20)])
def test_tiny_unet_vjp(inch, H, W, S):
    cfg = gcfg.tiny_unet(inch)
    net, sd = _unet(cfg)
    x = randn(2, inch, H, W, seed=1)
    t = torch.tensor([981, 17])
    ctx = randn(2, S, cfg.cross_attention_dim, seed=2)
    cot = randn(2, cfg.out_channels, H, W, seed=3)
    xr = x.clone().requires_grad_()
    ref_eps = M.unet_forward(sd, cfg, xr, t, ctx)
    (ref,) = torch.autograd.grad(ref_eps, xr, cot)
    xd = x.to(DEV).requires_grad_()
    eps = net(xd, t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample
    (got,) = torch.autograd.grad(eps, xd, cot.to(DEV))
    report(f"tiny unet vjp eps in{inch} {H}x{W}", eps.detach().cpu(), ref_eps.detach(), 3e-2)
    report(f"tiny unet vjp d_x in{inch} {H}x{W}", got.cpu(), ref, 5e-2)
    # the no-grad path is untouched and equal to the differentiable path's forward value
    with torch.no_grad():
        plain = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample
    assert torch.equal(plain, eps.detach())


def test_tiny_sdxl_style_unet_vjp():
    """linear projections, transformer depth 2, text_time conditioning"""
    cfg = gcfg.UNetConfig(block_out_channels=(32, 64, 128), attn_levels=(False, True, True), num_heads=(2, 2, 4),
                          transformer_depth=(1, 2, 2), cross_attention_dim=64, use_linear_projection=True, sample_size=16)
    net, sd = _unet(cfg)
    x = randn(2, 4, 16, 16, seed=1)
    t = torch.tensor([500, 40])
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=2)
    cot = randn(2, 4, 16, 16, seed=3)
    xr = x.clone().requires_grad_()
    (ref,) = torch.autograd.grad(M.unet_forward(sd, cfg, xr, t, ctx), xr, cot)
    xd = x.to(DEV).requires_grad_()
    (got,) = torch.autograd.grad(net(xd, t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample, xd, cot.to(DEV))
    report("tiny linear-proj depth-2 unet vjp", got.cpu(), ref, 5e-2)


@pytest.mark.parametrize("r", [8, 40, 1000])
def test_tiny_unet_vjp_with_tome(r):
    """Token merging in the reverse sweep: the merge's adjoint spreads d K / d V of a merged row over the tokens that went
    into it (divided by their count); the matching indices carry no gradient, as under autograd in the oracle."""
    cfg = gcfg.tiny_unet()
    net, sd = _unet(cfg)
    net.set_tome(r)
    x = randn(2, 4, 16, 16, seed=1)
    t = torch.tensor([981, 17])
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=2)
    cot = randn(2, 4, 16, 16, seed=3)
    xr = x.clone().requires_grad_()
    ref_eps = M.unet_forward(sd, cfg, xr, t, ctx, tome_r=r)
    (ref,) = torch.autograd.grad(ref_eps, xr, cot)
    xd = x.to(DEV).requires_grad_()
    eps = net(xd, t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample
    (got,) = torch.autograd.grad(eps, xd, cot.to(DEV))
    report(f"tiny unet tome r={r} eps", eps.detach().cpu(), ref_eps.detach(), 3e-2)
    report(f"tiny unet tome r={r} vjp d_x", got.cpu(), ref, 5e-2)
    net.set_tome(0)
    (plain,) = torch.autograd.grad(net(xd, t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample, xd, cot.to(DEV))
    assert float((plain - got).abs().max()) > 1e-4          # merging really changed the gradient


@pytest.mark.parametrize("r", [0, 40])
def test_reverse_sweep_of_a_sample_range_of_the_kept_batch(r):
    """gyre_unet_vjp_finish_range (modules.grad_samples): ONE activation-keeping pass over cat[uncond, cond] and the reverse sweep of
    the conditional half = the keeping pass + sweep of the conditional half alone (what the guided mode ran before) and the plain
    pass of the unconditional half; the other samples get a zero gradient.  With and without token merging (whose index arrays are
    laid out for the full batch)."""
    from gyre_amd.modules import grad_samples
    cfg = gcfg.tiny_unet()
    net, sd = _unet(cfg)
    net.set_tome(r)
    x1 = randn(2, 4, 16, 16, seed=1)
    x = torch.cat([x1, x1]).to(DEV)                       # the CFG layout: both halves carry the same latents
    t = torch.tensor([981, 17, 981, 17]).to(DEV)
    ctx = randn(4, 77, cfg.cross_attention_dim, seed=2).to(DEV)
    cot = randn(2, 4, 16, 16, seed=3).to(DEV)
    xa = x[2:].clone().requires_grad_()
    eps_a = net(xa, t[2:], encoder_hidden_states=ctx[2:].contiguous()).sample
    (dx_a,) = torch.autograd.grad(eps_a, xa, cot)
    with torch.no_grad():
        eps_u = net(x[:2], t[:2], encoder_hidden_states=ctx[:2].contiguous()).sample
    xb = x.clone().requires_grad_()
    with grad_samples(2, 2):
        eps_b = net(xb, t, encoder_hidden_states=ctx).sample
    (dx_b,) = torch.autograd.grad(eps_b[2:], xb, cot)
    report(f"range sweep tome {r}: eps of the differentiated half", eps_b[2:].detach().float().cpu(), eps_a.detach().float().cpu(), 2e-2)
    report(f"range sweep tome {r}: eps of the detached half", eps_b[:2].detach().float().cpu(), eps_u.float().cpu(), 2e-2)
    report(f"range sweep tome {r}: d_x", dx_b[2:].float().cpu(), dx_a.float().cpu(), 3e-2)
    assert float(dx_b[:2].abs().max()) == 0.0
    net.set_tome(0)


def test_tiny_vae_decode_vjp():
    cfg = gcfg.tiny_vae()
    sd = weights.synthetic_state_dict(weights.vae_param_shapes(cfg), 0)
    net = GyreHipVAE(cfg)
    net.load_state_dict(sd)
    net = net.to(DEV)
    z = randn(2, 4, 8, 12, seed=5)
    zr = z.clone().requires_grad_()
    ref_img = M.vae_decode(sd, cfg, zr)
    cot = randn(*ref_img.shape, seed=6)
    (ref,) = torch.autograd.grad(ref_img, zr, cot)
    zd = z.to(DEV).requires_grad_()
    img = net.decode(zd).sample
    (got,) = torch.autograd.grad(img, zd, cot.to(DEV))
    report("tiny vae decode (grad path) image", img.detach().cpu(), ref_img.detach(), 3e-2)
    report("tiny vae decode vjp d_z", got.cpu(), ref, 5e-2)


def test_sd15_unet_vjp_full_size():
    """Full SD1.5 UNet at 32x32 latents (the oracle's fp32 CPU backward takes about a minute at 64x64)."""
    cfg = gcfg.sd15_unet()
    net, sd = _unet(cfg)
    x = randn(1, 4, 32, 32, seed=7)
    t = torch.tensor([700])
    ctx = randn(1, 77, 768, seed=8)
    cot = randn(1, 4, 32, 32, seed=9)
    xr = x.clone().requires_grad_()
    (ref,) = torch.autograd.grad(M.unet_forward(sd, cfg, xr, t, ctx), xr, cot)
    xd = x.to(DEV).requires_grad_()
    (got,) = torch.autograd.grad(net(xd, t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample, xd, cot.to(DEV))
    report("SD1.5 unet vjp 1x4x32x32", got.cpu(), ref, 5e-2)
    # linearity in the cotangent: a property that holds at any size
    (g2,) = torch.autograd.grad(net(xd, t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample, xd, 2 * cot.to(DEV))
    report("SD1.5 unet vjp linear in the cotangent", g2.cpu(), 2 * got.cpu(), 1e-2)


def test_sd15_vae_decode_vjp_cutout_size():
    """The size the reference decodes CLIP cut-outs at: 224 / 8 = 28x28 latents (clipguided.py:127-131)."""
    cfg = gcfg.sd15_vae()
    sd = weights.synthetic_state_dict(weights.vae_param_shapes(cfg), 0)
    net = GyreHipVAE(cfg)
    net.load_state_dict(sd)
    net = net.to(DEV)
    z = randn(1, 4, 28, 28, seed=5)
    zr = z.clone().requires_grad_()
    ref_img = M.vae_decode(sd, cfg, zr)
    cot = randn(*ref_img.shape, seed=6)
    (ref,) = torch.autograd.grad(ref_img, zr, cot)
    zd = z.to(DEV).requires_grad_()
    (got,) = torch.autograd.grad(net.decode(zd).sample, zd, cot.to(DEV))
    report("SD1.5 vae decode vjp 28x28", got.cpu(), ref, 5e-2)


def test_tiny_sdxl_unet_vjp_with_text_time_and_bf16_io():
    """SDXL-style added conditioning enters the time embedding (no gradient path to the sample) and bf16 boundary tensors"""
    cfg = gcfg.tiny_sdxl_unet()
    net, sd = _unet(cfg)
    x = randn(2, 4, 16, 16, seed=1)
    t = torch.tensor([500, 40])
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=2)
    added = {"text_embeds": randn(2, 32, seed=4), "time_ids": torch.tensor([[128., 128, 0, 0, 128, 128]] * 2)}
    cot = randn(2, 4, 16, 16, seed=3)
    xr = x.clone().requires_grad_()
    (ref,) = torch.autograd.grad(M.unet_forward(sd, cfg, xr, t, ctx, added_cond=added), xr, cot)
    add_dev = {k: v.to(DEV) for k, v in added.items()}
    xd = x.to(DEV).requires_grad_()
    (got,) = torch.autograd.grad(net(xd, t.to(DEV), encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=add_dev).sample, xd, cot.to(DEV))
    report("tiny sdxl unet vjp (text_time)", got.cpu(), ref, 5e-2)
    xb = x.to(DEV, torch.bfloat16).requires_grad_()
    eps = net(xb, t.to(DEV), encoder_hidden_states=ctx.to(DEV, torch.bfloat16), added_cond_kwargs=add_dev).sample
    (gb,) = torch.autograd.grad(eps, xb, cot.to(DEV, torch.bfloat16))
    assert eps.dtype == torch.bfloat16 and gb.dtype == torch.bfloat16
    report("tiny sdxl unet vjp bf16 io", gb.float().cpu(), ref, 6e-2)


def test_vjp_batch_one_and_repeatable():
    cfg = gcfg.tiny_unet()
    net, sd = _unet(cfg)
    x = randn(1, 4, 12, 20, seed=1)
    ctx = randn(1, 33, cfg.cross_attention_dim, seed=2)
    cot = randn(1, 4, 12, 20, seed=3)
    xr = x.clone().requires_grad_()
    (ref,) = torch.autograd.grad(M.unet_forward(sd, cfg, xr, torch.tensor([7]), ctx), xr, cot)
    xd = x.to(DEV).requires_grad_()
    (a,) = torch.autograd.grad(net(xd, 7, encoder_hidden_states=ctx.to(DEV)).sample, xd, cot.to(DEV))
    (b,) = torch.autograd.grad(net(xd, 7, encoder_hidden_states=ctx.to(DEV)).sample, xd, cot.to(DEV))
    report("tiny unet vjp batch 1, 12x20, S=33", a.cpu(), ref, 5e-2)
    assert torch.equal(a, b)


def test_vjp_pending_state_is_dropped_by_other_calls_and_recomputed():
    """Two forwards before the first backward, and a no-grad forward in between: the pending begin / finish state is lost,
    backward falls back to the one-shot sweep - same gradient either way."""
    cfg = gcfg.tiny_unet()
    net, _ = _unet(cfg)
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=2).to(DEV)
    t = torch.tensor([500, 40], device=DEV)
    cot = randn(2, 4, 16, 16, seed=3).to(DEV)
    xa, xb = randn(2, 4, 16, 16, seed=1).to(DEV).requires_grad_(), randn(2, 4, 16, 16, seed=9).to(DEV).requires_grad_()
    (ga_direct,) = torch.autograd.grad(net(xa, t, encoder_hidden_states=ctx).sample, xa, cot)      # begin / finish pair
    ea = net(xa, t, encoder_hidden_states=ctx).sample
    eb = net(xb, t, encoder_hidden_states=ctx).sample                                               # overwrites a's state
    (ga,) = torch.autograd.grad(ea, xa, cot)                                                        # -> recomputed
    (gb,) = torch.autograd.grad(eb, xb, cot)                                                        # -> pending pair
    assert torch.equal(ga, ga_direct)
    ec = net(xb, t, encoder_hidden_states=ctx).sample
    with torch.no_grad():
        net(xa, t, encoder_hidden_states=ctx)                                                       # drops the native state
    (gc,) = torch.autograd.grad(ec, xb, cot)
    assert torch.equal(gb, gc)


def test_vjp_follows_weight_updates():
    """the reverse sweep's transposed weights are cached per handle: a re-upload (what a per-request LoRA merge does) must
    refresh them"""
    cfg = gcfg.tiny_unet()
    net, sd = _unet(cfg, seed=0)
    x = randn(2, 4, 16, 16, seed=1)
    t = torch.tensor([981, 17])
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=2)
    cot = randn(2, 4, 16, 16, seed=3)

    def both(state):
        xr = x.clone().requires_grad_()
        (ref,) = torch.autograd.grad(M.unet_forward(state, cfg, xr, t, ctx), xr, cot)
        xd = x.to(DEV).requires_grad_()
        (got,) = torch.autograd.grad(net(xd, t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample, xd, cot.to(DEV))
        return got.cpu(), ref
    g0, r0 = both(sd)
    report("vjp before the weight update", g0, r0, 5e-2)
    sd2 = weights.synthetic_state_dict(weights.unet_param_shapes(cfg), 7)
    net.load_state_dict(sd2)
    g1, r1 = both(sd2)
    report("vjp after the weight update", g1, r1, 5e-2)
    assert float((r1 - r0).norm() / r0.norm()) > 0.1          # the two weight sets really give different gradients
