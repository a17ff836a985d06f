"""CPU restatement of the scheduler / sampler / CFG / RNG arithmetic on the hot
loop.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Pinned by tests/golden/*.npz (made by tests/golden/make_golden.py, which imports
the reference's own modules in the build container) wherever the reference owns
the code; the k-diffusion pieces ([3P], submodule gyre/src/k-diffusion is empty
in the reference checkout) follow k-diffusion's published algorithm and are
cross-checked against the reference's in-tree witnesses
(gyre/pipeline/kschedulers/scheduling_utils.py:107-126,
 gyre/pipeline/kschedulers/scheduling_euler_ancestral_discrete.py:108-157).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

Tensor = torch.Tensor


# ---- per-image RNG: gyre/pipeline/randtools.py:11-64 ------------------------
def batched_randn(shape: Sequence[int], generators: List[torch.Generator], dtype=torch.float32) -> Tensor:
    if shape[0] % len(generators) != 0:
        raise ValueError(f"shape[0] ({shape[0]}) needs to be a multiple of len(generators) ({len(generators)})")
    return torch.cat([torch.randn((1, *shape[1:]), generator=g, device=g.device, dtype=dtype)
                      for g in generators * (shape[0] // len(generators))], dim=0)


def batched_rand(shape: Sequence[int], generators: List[torch.Generator], dtype=torch.float32) -> Tensor:
    if shape[0] % len(generators) != 0:
        raise ValueError(f"shape[0] ({shape[0]}) needs to be a multiple of len(generators) ({len(generators)})")
    return torch.cat([torch.rand((1, *shape[1:]), generator=g, device=g.device, dtype=dtype)
                      for g in generators * (shape[0] // len(generators))], dim=0)


# ---- noise schedule: gyre/pipeline/common_scheduler.py:410-428 --------------
def get_betas(n: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> Tensor:
    return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n) ** 2


def get_alphas_cumprod(betas: Tensor) -> Tensor:
    return torch.cumprod(1.0 - betas, dim=0)


class DiscreteScheduleRef:
    """k_diffusion.external.DiscreteSchedule [3P] (quantize=True as at
    common_scheduler.py:344); witness kschedulers/scheduling_utils.py:107-126."""

    def __init__(self, alphas_cumprod: Optional[Tensor] = None):
        if alphas_cumprod is None:
            alphas_cumprod = get_alphas_cumprod(get_betas())
        self.sigmas = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5
        self.log_sigmas = self.sigmas.log()

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def sigma_to_t(self, sigma: Tensor) -> Tensor:
        log_sigma = sigma.log()
        dists = log_sigma - self.log_sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def t_to_sigma(self, t: Tensor) -> Tensor:
        t = t.float()
        low, high, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[low] + w * self.log_sigmas[high]).exp()

    # common_scheduler.py:476-514 (no sigma_min/max overrides)
    def get_sigmas(self, n: int) -> Tensor:
        t = torch.linspace(len(self.sigmas) - 1, 0, n)
        return torch.cat([self.t_to_sigma(t), torch.zeros(1)])

    # k_diffusion.sampling.get_sigmas_karras [3P], called at common_scheduler.py:488-500
    def get_sigmas_karras(self, n: int, rho: float = 7.0) -> Tensor:
        ramp = torch.linspace(0, 1, n)
        min_inv = float(self.sigma_min) ** (1 / rho)
        max_inv = float(self.sigma_max) ** (1 / rho)
        sig = (max_inv + ramp * (min_inv - max_inv)) ** rho
        return torch.cat([sig, torch.zeros(1)])


class EpsDenoiserRef:
    """k_diffusion.external.DiscreteEpsDDPMDenoiser.forward [3P] as subclassed at
    common_scheduler.py:342-347: x0 = x + eps(x*c_in, t(sigma)) * c_out."""

    def __init__(self, eps_model: Callable[[Tensor, Tensor], Tensor], schedule: DiscreteScheduleRef):
        self.inner = eps_model
        self.s = schedule

    def __call__(self, x: Tensor, sigma: Tensor) -> Tensor:
        c_out = -sigma
        c_in = 1.0 / (sigma ** 2 + 1.0) ** 0.5
        t = self.s.sigma_to_t(sigma)
        shape = (-1,) + (1,) * (x.ndim - 1)
        eps = self.inner(x * c_in.view(shape), t)
        return x + eps * c_out.view(shape)


# ---- CFG: gyre/pipeline/unet/cfg.py:27-57 -----------------------------------
def cfg_parallel(unet_f: Callable[[Tensor, Tensor], Tensor], guidance_scale: float):
    def call(latents: Tensor, t: Tensor) -> Tensor:
        latents = torch.cat([latents, latents])
        if isinstance(t, torch.Tensor) and t.shape:
            t = torch.cat([t, t])
        u, g = unet_f(latents, t).chunk(2)
        return u + guidance_scale * (g - u)
    return call


def cfg_sequential(unet_g, unet_u, guidance_scale: float):
    def call(latents: Tensor, t: Tensor) -> Tensor:
        g = unet_g(latents, t)
        u = unet_u(latents, t)
        return u + guidance_scale * (g - u)
    return call


# ---- samplers ----------------------------------------------------------------
def sample_dpmpp_2m(model, x: Tensor, sigmas: Tensor, warmup_lms: bool = False, ddim_cutoff: float = 0.0,
                    callback=None) -> Tensor:
    """gyre/pipeline/schedulers/sample_dpmpp_2m.py:6-50 (bound with warmup_lms=True,
    ddim_cutoff=0.1 at gyre/pipeline/samplers.py:58-66)."""
    s_in = x.new_ones([x.shape[0]])
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    old_denoised = None
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "denoised": denoised})
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        h = t_next - t
        if old_denoised is None and warmup_lms:
            r = 1 / 2
            s = t + r * h
            x_2 = (sigma_fn(s) / sigma_fn(t)) * x - (-h * r).expm1() * denoised
            denoised_i = model(x_2, sigma_fn(s) * s_in)
        elif sigmas[i + 1] <= ddim_cutoff or old_denoised is None:
            denoised_i = denoised
        else:
            h_last = t - t_fn(sigmas[i - 1])
            r = h_last / h
            denoised_i = (1 + 1 / (2 * r)) * denoised - (1 / (2 * r)) * old_denoised
        x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * denoised_i
        old_denoised = denoised
    return x


def get_ancestral_step(sigma_from: Tensor, sigma_to: Tensor, eta: float = 1.0):
    """k_diffusion.sampling.get_ancestral_step [3P]; witness
    kschedulers/scheduling_euler_ancestral_discrete.py:137-140 (eta=1)."""
    if not eta:
        return sigma_to, torch.zeros_like(sigma_to)
    sigma_up = torch.minimum(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def sample_euler_ancestral(model, x: Tensor, sigmas: Tensor, noise_sampler, eta: float = 1.0,
                           s_noise: float = 1.0) -> Tensor:
    """k_diffusion.sampling.sample_euler_ancestral [3P] (selected at
    gyre/pipeline/samplers.py:50); witness
    kschedulers/scheduling_euler_ancestral_discrete.py:108-157."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta)
        d = (x - denoised) / sigmas[i]
        dt = sigma_down - sigmas[i]
        x = x + d * dt
        if sigmas[i + 1] > 0:
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * s_noise * sigma_up
    return x


def sample_euler(model, x: Tensor, sigmas: Tensor) -> Tensor:
    """k_diffusion.sampling.sample_euler [3P] with s_churn=0 (samplers.py:49)."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        d = (x - denoised) / sigmas[i]
        x = x + d * (sigmas[i + 1] - sigmas[i])
    return x


# ---- Txt2imgMode.generateLatents: gyre/pipeline/unified_pipeline.py:193-237 --
def txt2img_latents(generators: List[torch.Generator], channels: int, lat_h: int, lat_w: int,
                    unet_sample_size: int, sigma0: Tensor, dtype=torch.float32) -> Tensor:
    B = len(generators)
    shape = (B, channels, lat_h, lat_w)
    mid = batched_randn([B, channels, unet_sample_size, unet_sample_size], generators, dtype)
    off2 = (unet_sample_size - lat_h) // 2
    off3 = (unet_sample_size - lat_w) // 2
    if off2 > 0:
        mid = mid[:, :, off2:off2 + lat_h, :]
    if off3 > 0:
        mid = mid[:, :, :, off3:off3 + lat_w]
    if off2 >= 0 and off3 >= 0:
        latents = mid
    else:
        latents = batched_randn(shape, generators, dtype)
        o2 = (latents.shape[2] - mid.shape[2]) // 2
        o3 = (latents.shape[3] - mid.shape[3]) // 2
        latents[:, :, o2:o2 + mid.shape[2], o3:o3 + mid.shape[3]] = mid
    return latents * sigma0  # prepare_initial_latents, common_scheduler.py:540


# ---- mask helpers: gyre/pipeline/unified_pipeline.py:340-395 -----------------
def downscale_boxop_1d(inp: Tensor, scale: int = 8, op: str = "max") -> Tensor:
    shape = inp.shape[:-1] + (inp.shape[-1] // scale, scale)
    return getattr(inp.reshape(shape), op)(dim=-1).values


def downscale_boxop_2d(inp: Tensor, scale: int = 8, op: str = "max") -> Tensor:
    mid = downscale_boxop_1d(inp, scale, op)
    return downscale_boxop_1d(mid.transpose(-2, -1), scale, op).transpose(-2, -1)


def mask_to_latent_mask(mask: Tensor, inputIs1K0D: bool = True) -> Tensor:
    mask = downscale_boxop_2d(mask, 8, "min" if inputIs1K0D else "max")
    return mask[:, [0, 0, 0, 0]]


def round_mask(mask: Tensor, threshold: float = 0.5) -> Tensor:
    mask = mask.clone()
    mask[mask >= threshold] = 1
    mask[mask < 1] = 0
    return mask


# ------------------------------------------------------------------------------
# DPM-Solver++ multistep, orders 1-3 (reference samplers.py:34-45 -> diffusers DPMSolverMultistepScheduler, [3P]
# ~=0.16.0, not vendored: PARITY UNPINNED; restated from Lu et al. 2022, "DPM-Solver++", Alg. 2 and eq. (12)-(13)
# in data-prediction form with the diffusers defaults: midpoint 2nd-order form, lower_order_final for < 15 steps,
# timesteps = round(linspace(0, 999, n+1))[::-1][:-1], last step to t = 0).
# Written on tensors of lambda / alpha / sigma so that it shares no code with the product's scalar version.
# ------------------------------------------------------------------------------
def dpmsolverpp_multistep_ref(eps_model, x: Tensor, n: int, order: int, alphas_cumprod: Optional[Tensor] = None,
                              start: int = 0) -> Tensor:
    import numpy as np
    ac = (get_alphas_cumprod(get_betas()) if alphas_cumprod is None else alphas_cumprod).double()
    alpha, sigma = ac.sqrt(), (1 - ac).sqrt()
    lam = alpha.log() - sigma.log()
    ts = [int(v) for v in np.linspace(0, len(ac) - 1, n + 1).round()[::-1][:-1]]
    hist = []            # (timestep, x0 prediction), newest last
    warm = 0
    for i in range(start, len(ts)):
        s0 = ts[i]
        t = ts[i + 1] if i + 1 < len(ts) else 0
        x0 = (x - sigma[s0] * eps_model(x, s0)) / alpha[s0]
        hist.append((s0, x0))
        h = lam[t] - lam[s0]
        phi = torch.expm1(-h)
        k = min(order, warm + 1)
        if len(ts) < 15:
            if i == len(ts) - 1:
                k = 1
            elif i == len(ts) - 2:
                k = min(k, 2)
        new = sigma[t] / sigma[s0] * x - alpha[t] * phi * x0
        if k >= 2:
            s1, m1 = hist[-2]
            r0 = (lam[s0] - lam[s1]) / h
            d10 = (x0 - m1) / r0
            if k == 2:
                new = new - 0.5 * alpha[t] * phi * d10
            else:
                s2, m2 = hist[-3]
                r1 = (lam[s1] - lam[s2]) / h
                d11 = (m1 - m2) / r1
                d1 = d10 + r0 / (r0 + r1) * (d10 - d11)
                d2 = (d10 - d11) / (r0 + r1)
                new = new + alpha[t] * (phi / h + 1.0) * d1 - alpha[t] * ((phi + h) / h ** 2 - 0.5) * d2
        x = new.to(x.dtype)
        warm = min(warm + 1, order)
    return x


# ------------------------------------------------------------------------------
# EnhancedInpaintMode._fillWithShapedNoise, noise_mode 5 (reference unified_pipeline.py:466-601; lmask_mode 3 = high
# mask).  Pinned by tests/golden shaped_noise_* (generated from the reference itself).
# ------------------------------------------------------------------------------
def fill_with_shaped_noise_ref(init_latents: Tensor, latent_mask: Tensor, generators, shaped_noise_strength: float) -> Tensor:
    import numpy as np
    high = round_mask(latent_mask, 0.001)
    masked = init_latents * high
    donor = high[0, 0] >= 0.5
    filled = torch.empty_like(init_latents)
    for b, g in enumerate(generators):
        npseed = torch.randint(low=0, high=torch.iinfo(torch.int32).max, size=[1], generator=g, dtype=torch.int32)
        rng = np.random.default_rng(npseed.numpy())
        shuffled = torch.empty_like(masked[b])
        for c in range(masked.shape[1]):
            pool = masked[b, c][donor].numpy()
            shuffled[c] = torch.from_numpy(rng.choice(pool, (1, 1) + tuple(masked.shape[2:])))[0, 0]
        white = torch.zeros_like(masked[b:b + 1]).normal_(generator=g)[0]
        filled[b] = white * (1 - shaped_noise_strength) + shuffled * shaped_noise_strength
    return init_latents * latent_mask + filled * (1 - latent_mask)
