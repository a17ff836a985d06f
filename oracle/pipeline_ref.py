"""End-to-end CPU restatement of the reference's txt2img / img2img call path, fp32.
TEST INFRASTRUCTURE and the reported CPU baseline (see oracle/__init__.py).

Follows gyre/pipeline/unified_pipeline.py:1912-2531 with the k-diffusion
sampler path: embeddings -> CFGUNet_Parallel (unet/cfg.py:41-57) -> DiscreteEpsDDPMDenoiser
(common_scheduler.py:342-347) -> sampler (samplers.py:47-67) -> vae.decode(latents/0.18215)
-> (x/2+0.5).clamp(0,1)  (unified_pipeline.py:2488-2491).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import models_ref as M
from . import sched_ref as S

Tensor = torch.Tensor


def generate_ref(unet_sd, unet_cfg, vae_sd, vae_cfg, text_emb: Tensor, uncond_emb: Optional[Tensor],
                 seeds: Sequence[int], height: int = 512, width: int = 512, steps: int = 20,
                 guidance_scale: float = 7.5, sampler: str = "euler_a", image: Optional[Tensor] = None,
                 strength: float = 0.8, decode: bool = True, unet_sample_size: int = 64):
    B = len(seeds)
    gens = [torch.Generator().manual_seed(int(s)) for s in seeds]
    sch = S.DiscreteScheduleRef()
    sigmas = sch.get_sigmas(steps)
    evals = [0]

    def unet_f(latents, t):
        evals[0] += 1
        ctx = torch.cat([uncond_emb, text_emb]) if guidance_scale > 1 else text_emb
        return M.unet_forward(unet_sd, unet_cfg, latents, t, ctx)

    eps = S.cfg_parallel(unet_f, guidance_scale) if guidance_scale > 1 else unet_f
    den = S.EpsDenoiserRef(eps, sch)
    lat_h, lat_w = height // 8, width // 8
    start = 0
    if image is None:
        x = S.txt2img_latents(gens, 4, lat_h, lat_w, unet_sample_size, sigmas[0])
    else:  # Img2imgMode, unified_pipeline.py:283-337
        img = 2.0 * image[:, [0, 1, 2]] - 1.0
        mom = M.vae_encode_moments(vae_sd, vae_cfg, img)
        lat = torch.cat([M.vae_posterior_sample(mom, g) for g in gens], dim=0) * 0.18215
        start = max(steps - min(int(steps * strength), steps), 0)
        noise = S.batched_randn(lat.shape, gens)
        sig = sch.t_to_sigma(sch.sigma_to_t(sigmas[start]))
        x = lat + noise * sig
    sig = sigmas[start:]
    if sampler == "euler_a":
        x = S.sample_euler_ancestral(den, x, sig, lambda a, b: S.batched_randn(x.shape, gens))
    elif sampler == "euler":
        x = S.sample_euler(den, x, sig)
    elif sampler == "dpmpp_2m":
        x = S.sample_dpmpp_2m(den, x, sig, warmup_lms=True, ddim_cutoff=0.1)
    else:
        raise NotImplementedError(sampler)
    if not decode:
        return x, evals[0]
    img = M.vae_decode(vae_sd, vae_cfg, x / 0.18215)
    return (img / 2 + 0.5).clamp(0, 1), evals[0]


def psnr(a: Tensor, b: Tensor, peak: float = 1.0) -> float:
    mse = float(((a.double() - b.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * torch.log10(torch.tensor(peak * peak / mse)).item()
