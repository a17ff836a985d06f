"""Build-container probe: the REFERENCE's ClipGuidedMode (gyre/pipeline/unet/clipguided.py, imported from /root/reference)
against gyre_amd.clipguided.ClipGuidedMode on the same toy differentiable UNet / VAE / tiny CLIP, same generators.

Absent third-party pieces get functional stand-ins: torchvision.transforms.Normalize / Resize, k_diffusion.utils.append_dims,
and the un-vendored ResizeRight submodule (routed to gyre_amd.resize.resize_right, so cut-out RESAMPLING is common to both
sides - everything around it is the reference's own code: draw order, crops, the view / stack of the cut-outs, loss, loss
history, flat-loss stop, guided / mixed stems, the k-diffusion and diffusers-style corrections).
Prints one JSON object.  Run by tests/test_reference_clipguided.py; nothing here ships."""
import json
import os
import sys
from types import SimpleNamespace

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def install_reference():
    import make_golden as mg
    mg._install()
    sys.path.insert(0, mg.REF)
    import torchvision.transforms as T                      # stub module: give it the two transforms the mode uses

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            m = torch.as_tensor(self.mean, dtype=x.dtype).view(1, -1, 1, 1)
            s = torch.as_tensor(self.std, dtype=x.dtype).view(1, -1, 1, 1)
            return (x - m) / s

    class Resize:
        def __init__(self, size):
            self.size = size

        def __call__(self, x):
            from gyre_amd.clipguided import _resize_short_edge
            return _resize_short_edge(x, self.size)
    T.Normalize, T.Resize = Normalize, Resize
    import k_diffusion.utils as ku
    ku.append_dims = lambda x, n: x[(...,) + (None,) * (n - x.ndim)]
    import gyre.resize_right as rr
    from gyre_amd.resize import resize_right as mine

    def resize(input, scale_factors=None, out_shape=None, interp_method=None, support_sz=None, antialiasing=True,
               by_convs=False, scale_tolerance=None, max_numerator=10, pad_mode="constant"):
        return mine(input, out_shape=out_shape, scale_factors=scale_factors, antialiasing=antialiasing, pad_mode=pad_mode)
    rr.resize_right.resize = resize
    from gyre.pipeline.unet import clipguided as ref
    from gyre.pipeline.unet.cfg import CFGChildUnets, CFGUNet_Parallel
    from gyre.pipeline import common_scheduler as cs
    return ref, CFGChildUnets, CFGUNet_Parallel, cs


def toys(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn(4, 4, 3, 3, generator=g) * 0.2
    w2 = torch.randn(4, 4, 3, 3, generator=g) * 0.2
    wd = torch.randn(3, 4, 3, 3, generator=g) * 0.3
    conv = torch.nn.functional.conv2d

    def unet_g(x, t):
        return torch.tanh(conv(x, w1, padding=1)) * (1 + 0.001 * float(t))

    def unet_u(x, t):
        return torch.tanh(conv(x, w2, padding=1)) * 0.5

    def vae_decode(z):                                      # differentiable 8x "decoder" with values around [-1, 1]
        up = torch.nn.functional.interpolate(z, scale_factor=8, mode="nearest")
        return torch.tanh(conv(up, wd, padding=1))
    from transformers import CLIPConfig, CLIPModel
    torch.manual_seed(1)
    clip = CLIPModel(CLIPConfig(text_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                                 vocab_size=1000, max_position_embeddings=16, bos_token_id=1, eos_token_id=2),
                                vision_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                                   num_attention_heads=2, image_size=32, patch_size=8),
                                projection_dim=16)).eval()
    for p in clip.parameters():
        p.requires_grad_(False)
    from gyre_amd.clipguided import _features
    clip_t = SimpleNamespace(get_image_features=lambda im: _features(clip.get_image_features(im)))
    temb = torch.nn.functional.normalize(torch.randn(B, 16, generator=g), dim=-1)
    return unet_g, unet_u, vae_decode, clip_t, temb


SIGMAS = [9.0, 6.1, 3.9, 2.2, 1.4, 0.8]
TS = [901, 741, 581, 421, 261, 101]


def run_case(ref_mods, case, only=None):
    ref, CFGChildUnets, CFGUNet_Parallel, cs = ref_mods
    from gyre_amd import clipguided as mine
    from gyre_amd import schedulers as S
    B, gs = 2, 5.0
    unet_g, unet_u, vae_decode, clip_t, temb = toys(B)
    fe = SimpleNamespace(image_mean=[0.48, 0.45, 0.40], image_std=[0.26, 0.26, 0.27], size=32)
    cfg_kw = dict(guidance_scale=case.get("scale", 0.7), guidance_base=case.get("base", "guided"),
                  gradient_length=case.get("glen", 15), gradient_threshold=case.get("gthr", 0.01),
                  gradient_maxloss=case.get("gmax", 1.0), vae_cutouts=case.get("vae", 2), approx_cutouts=case.get("approx", 2),
                  no_cutouts=case.get("noc", False))
    x0 = torch.randn(B, 4, 8, 8, generator=torch.Generator().manual_seed(5)) * SIGMAS[0]
    out = {}

    def k_child_from(eps):                                   # KDiffusionUNetWrapper arithmetic
        def k(x, sigma, u=None):
            s = float(sigma)
            i = min(range(len(SIGMAS)), key=lambda j: abs(SIGMAS[j] - s))
            return x + eps(x * (1.0 / (s * s + 1) ** 0.5), TS[i]) * (-s)
        return k

    # ---------------- reference ----------------
    gens = [torch.Generator().manual_seed(100 + i) for i in range(B)]
    pipeline = SimpleNamespace(feature_extractor=fe, vae_scale_factor=8, execution_device="cpu", vae_decode=vae_decode,
                               clip_model=clip_t)
    cfg_children = CFGChildUnets(g=unet_g, u=unet_u, f=lambda x, t: torch.cat([unet_u(x.chunk(2)[0], t), unet_g(x.chunk(2)[1], t)]))
    if case["kind"] == "k":
        sched = SimpleNamespace()
    else:
        sched = object.__new__(cs.DiffusersScheduler)
        sched.scheduler = SimpleNamespace(alphas_cumprod=S.DiscreteSchedule().alphas_cumprod)
    wrapped_mode = SimpleNamespace(wrap_guidance_unet=lambda c, s_, b: CFGUNet_Parallel(c, s_, b), wrap_k_unet=lambda u: u,
                                   wrap_d_unet=lambda u: u, wrap_unet=lambda u: u, generateLatents=lambda: None)
    rmode = ref.ClipGuidedMode(wrapped_mode=wrapped_mode, scheduler=sched, pipeline=pipeline, dtype=torch.float32,
                               text_embeddings_clip=temb, config=ref.ClipGuidanceConfig(**cfg_kw), generators=gens,
                               reversible_ctx=__import__("contextlib").nullcontext)
    r_eps = rmode.wrap_guidance_unet(cfg_children, gs, B)
    r_outs = []
    x = x0.clone()
    if only == "mine":
        pass
    elif case["kind"] == "k":
        r_k = rmode.wrap_k_unet(k_child_from(r_eps))
        for s in SIGMAS:
            d = r_k(x, torch.tensor(s), u=0.5).detach()
            r_outs.append(d)
            x = d + (x - d) * 0.7
    else:
        for t in TS:
            e = r_eps(x, t).detach()
            r_outs.append(e)
            x = x - 0.1 * e
    # ---------------- ours ----------------
    gens = [torch.Generator().manual_seed(100 + i) for i in range(B)]
    msched = object.__new__(S.DiffusersScheduler) if case["kind"] == "d" else SimpleNamespace()
    if case["kind"] == "d":
        msched.sched = SimpleNamespace(alphas_cumprod=S.DiscreteSchedule().alphas_cumprod)
    mmode = mine.ClipGuidedMode(scheduler=msched, clip_model=clip_t, image_mean=fe.image_mean, image_std=fe.image_std,
                                clip_size=fe.size, vae_decode=vae_decode, vae_scale_factor=8, text_embeddings_clip=temb,
                                config=mine.ClipGuidanceConfig(**cfg_kw), generators=gens)
    child = S.CFGUNet_Parallel(lambda x2, t: torch.cat([unet_u(x2.chunk(2)[0], t), unet_g(x2.chunk(2)[1], t)]), gs, B)
    m_eps = mmode.wrap_guidance_unet(unet_g, unet_u, child, gs)
    m_outs = []
    x = x0.clone()
    if only is not None:
        r_outs = r_outs or [x0]
    with torch.no_grad():
        if only == "ref":
            return out
        if case["kind"] == "k":
            kc = k_child_from(m_eps)
            m_k = mmode.wrap_k_unet(lambda xx, sg, u: kc(xx, sg))
            for s in SIGMAS:
                d = m_k(x, torch.tensor(s), 0.5).detach()
                m_outs.append(d)
                x = d + (x - d) * 0.7
        else:
            for t in TS:
                e = m_eps(x, t).detach()
                m_outs.append(e)
                x = x - 0.1 * e
    if only is not None:
        return out
    out["max_abs_diff"] = max(float((a - b).abs().max()) for a, b in zip(r_outs, m_outs))
    out["ref_absmax"] = max(float(a.abs().max()) for a in r_outs)
    out["lossavg_ref"] = [round(v, 6) for v in rmode.lossavg]
    out["lossavg_diff"] = max([abs(a - b) for a, b in zip(rmode.lossavg, mmode.lossavg)] + [0.0])
    out["n_loss"] = [len(rmode.lossavg), len(mmode.lossavg)]
    out["flat"] = [bool(rmode.flatloss), bool(mmode.flatloss)]
    out["gen_state_equal"] = all(torch.equal(a.get_state(), b.get_state()) for a, b in zip(rmode.generators, mmode.generators))
    # guidance really acts: the guided output differs from the plain CFG output
    plain = k_child_from(child)(x0, SIGMAS[0]) if case["kind"] == "k" else child(x0, TS[0])
    out["guidance_effect"] = float((r_outs[0] - plain).abs().max())
    return out


CASES = {
    "k_guided_default": dict(kind="k"),
    "k_mixed_default": dict(kind="k", base="mixed"),
    "k_vae_only": dict(kind="k", vae=3, approx=0),
    "k_approx_only": dict(kind="k", vae=0, approx=3),
    "k_no_cutouts_vae": dict(kind="k", vae=0, approx=0, noc=True),
    "k_no_cutouts_approx": dict(kind="k", vae=0, approx=0, noc="approx"),
    "k_flatloss_stops": dict(kind="k", glen=2, gthr=10.0, gmax=100.0),
    "d_guided_default": dict(kind="d"),
    "d_mixed_vae_only": dict(kind="d", base="mixed", vae=2, approx=0),
}


def main():
    mods = install_reference()
    res = {name: run_case(mods, case) for name, case in CASES.items()}
    # no_cutouts=True while cut-outs are still requested: the reference compares B*cutouts image embeddings with B text
    # embeddings (clipguided.py:406-409) and fails for B > 1; same here
    for side in ("ref", "mine"):
        try:
            run_case(mods, dict(kind="k", noc=True), only=side)
            res["noc_with_cutouts_" + side] = "ok"
        except RuntimeError:
            res["noc_with_cutouts_" + side] = "RuntimeError"
    print("PROBE_JSON " + json.dumps(res))


if __name__ == "__main__":
    main()
