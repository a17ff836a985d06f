"""Host-side scheduler / sampler / CFG / RNG code (thin PyTorch, per north_star).

Mirrors the reference's own interfaces for this part of the path so that the
parity tests read like the reference's code:

  batched_randn / batched_rand    gyre/pipeline/randtools.py:11-64
  CFGUNet_Parallel / _Sequential  gyre/pipeline/unet/cfg.py:27-57
  KDiffusionUNetWrapper           gyre/pipeline/common_scheduler.py:342-347 (k-diffusion
                                  DiscreteEpsDDPMDenoiser, quantize=True)
  KDiffusionScheduler             gyre/pipeline/common_scheduler.py:392-623
  sample_dpmpp_2m                 gyre/pipeline/schedulers/sample_dpmpp_2m.py:6-50
  k-diffusion samplers            selected at gyre/pipeline/samplers.py:47-67

MI355X-first difference: the sigma schedule, sigma->t quantisation and every per-step
coefficient are computed on the HOST in fp32 (tiny tables) and applied to the device
latents as scalars, so the denoising loop never reads a device value back - no
host<->device sync per step; the UNet launches of consecutive steps queue back to back.
Latents stay fp32 between steps (the reference keeps them in the text-embedding dtype).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch

Tensor = torch.Tensor


# ------------------------------------------------------------------------------
# per-image RNG (batch-composition independent noise)
# ------------------------------------------------------------------------------
def batched_rand(shape: Sequence[int], generators: List[torch.Generator], device, dtype) -> Tensor:
    if shape[0] % len(generators) != 0:
        raise ValueError(f"shape[0] ({shape[0]}) needs to be a multiple of len(generators) ({len(generators)})")
    out = torch.cat([torch.rand((1, *shape[1:]), generator=g, device=g.device, dtype=dtype)
                     for g in generators * (shape[0] // len(generators))], dim=0)
    return out.to(device)


def batched_randn(shape: Sequence[int], generators: List[torch.Generator], device, dtype) -> Tensor:
    if shape[0] % len(generators) != 0:
        raise ValueError(f"shape[0] ({shape[0]}) needs to be a multiple of len(generators) ({len(generators)})")
    out = torch.cat([torch.randn((1, *shape[1:]), generator=g, device=g.device, dtype=dtype)
                     for g in generators * (shape[0] // len(generators))], dim=0)
    return out.to(device)


class BrownianTreeNoiseSampler:
    """`scheduler_noise_type = "brownian"` (reference common_scheduler.py:596-606: one seed per image drawn from that image's
    generator, then [3P k_diffusion.sampling.BrownianTreeNoiseSampler] over [3P torchsde.BrownianTree]): the noise of a step
    (sigma -> sigma_next) is the increment of ONE Brownian path per image over [sigma, sigma_next], divided by
    sqrt(|sigma_next - sigma|) - unit variance per step, but all steps (and all step counts) read the same path, which is
    what makes SDE samplers converge as the step count grows.

    PARITY UNPINNED: torchsde is not in the reference tree nor in this image, so the values of its tree (its entropy
    spawning and bridge order) cannot be reproduced; the CONSTRUCTION is the published one - a virtual Brownian tree
    (Li et al. 2020, "Scalable gradients for stochastic differential equations", the algorithm torchsde's BrownianTree
    implements): W(t0) = 0, W(t1) ~ N(0, t1 - t0), and W at a query point by repeated bisection with the Brownian bridge
    W(mid) = (W(a) + W(b)) / 2 + sqrt((b - a) / 4) * z(node), where z(node) is drawn from a counter-based stream keyed
    by (image seed, tree level, node index) - so any query order gives the same path - down to intervals of
    (t1 - t0) / 2^DEPTH, inside which the path is interpolated linearly.  Per-image trees: the noise of an image does not
    depend on what else is in the batch (the property the reference gets from per-image generators)."""

    DEPTH = 20

    def __init__(self, x: Tensor, sigma_min, sigma_max, seed: Sequence[int], transform=lambda v: v):
        self.transform = transform
        t0, t1 = float(transform(torch.as_tensor(sigma_min))), float(transform(torch.as_tensor(sigma_max)))
        self.t0, self.t1 = (t0, t1) if t0 <= t1 else (t1, t0)
        if x.shape[0] % len(seed) != 0:
            raise ValueError(f"shape[0] ({x.shape[0]}) needs to be a multiple of len(seed) ({len(seed)})")
        self.seeds = [int(s) for s in seed] * (x.shape[0] // len(seed))
        self.shape, self.device, self.dtype = tuple(x.shape[1:]), x.device, x.dtype
        self._gen = torch.Generator(device="cpu")
        self._w_cache = {}                                       # (image, node) -> W at that node: steps share their end points

    def _z(self, img: int, level: int, index: int) -> Tensor:
        # counter-based: splitmix64 of (seed, level, index) -> the seed of a one-shot generator
        v = (self.seeds[img] + 0x9E3779B97F4A7C15 * (level + 1) + 0xBF58476D1CE4E5B9 * (index + 1)) & 0xFFFFFFFFFFFFFFFF
        v = ((v ^ (v >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        v = ((v ^ (v >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        v ^= v >> 31
        self._gen.manual_seed(v & 0x7FFFFFFFFFFFFFFF)
        return torch.randn(self.shape, generator=self._gen, dtype=torch.float32)

    def _w(self, img: int, t: float) -> Tensor:
        """W(t) - W(t0) of image `img` (fp32, host)."""
        span = self.t1 - self.t0
        u = min(max((t - self.t0) / span, 0.0), 1.0) if span > 0 else 0.0
        key = (img, round(u * (1 << 40)))
        hit = self._w_cache.get(key)
        if hit is not None:
            return hit
        wa = torch.zeros(self.shape, dtype=torch.float32)
        wb = self._z(img, 0, 0) * span ** 0.5
        a, b, index = 0.0, 1.0, 0
        for level in range(1, self.DEPTH + 1):
            if u == a or u == b:
                break
            mid = 0.5 * (a + b)
            wm = 0.5 * (wa + wb) + self._z(img, level, index) * (0.25 * (b - a) * span) ** 0.5
            if u < mid:
                b, wb, index = mid, wm, 2 * index
            else:
                a, wa, index = mid, wm, 2 * index + 1
        w = wa if u == a else wb if u == b else wa + (wb - wa) * ((u - a) / (b - a))
        if len(self._w_cache) > 4 * len(self.seeds):
            self._w_cache.clear()
        self._w_cache[key] = w
        return w

    def __call__(self, sigma, sigma_next) -> Tensor:
        """(W(sigma_next) - W(sigma)) / sqrt|sigma_next - sigma|: the SIGNED increment, as k-diffusion returns it (its sort() takes the
        end points in ascending order for torchsde and multiplies the sign back in), so a reversed interval negates the noise.
        End points outside the tree's [t0, t1] are clamped to it (k-diffusion builds the tree over the positive sigmas only and
        asks for noise only in front of a positive sigma_next) and the increment is normalised by the CLAMPED interval, so the
        result keeps unit variance."""
        ta, tb = float(self.transform(torch.as_tensor(sigma))), float(self.transform(torch.as_tensor(sigma_next)))
        ta, tb = min(max(ta, self.t0), self.t1), min(max(tb, self.t0), self.t1)
        if ta == tb:
            raise ValueError("BrownianTreeNoiseSampler: empty interval")
        scale = 1.0 / abs(tb - ta) ** 0.5
        out = torch.stack([(self._w(i, tb) - self._w(i, ta)) * scale for i in range(len(self.seeds))])
        return out.to(self.device, self.dtype)


# ------------------------------------------------------------------------------
# classifier-free guidance wrappers
# ------------------------------------------------------------------------------
class CFGUNet_Parallel:
    """One UNet call on cat[x, x] with cat[uncond, cond] embeddings; eps = u + s*(g-u)."""

    def __init__(self, unet_f: Callable[[Tensor, object], Tensor], guidance_scale: float, batch_total: int):
        self.f, self.guidance_scale, self.batch_total = unet_f, guidance_scale, batch_total

    def __call__(self, latents: Tensor, t) -> Tensor:
        latents = torch.cat([latents, latents])
        if isinstance(t, torch.Tensor) and t.shape:
            t = torch.cat([t, t])
        # both halves carry the same latents and timesteps: the native UNet shares what they have in common
        from .modules import cfg_pairs
        with cfg_pairs():
            noise_pred = self.f(latents, t)
        u, g = noise_pred.chunk(2)
        return u + self.guidance_scale * (g - u)


class CFGUNet_Sequential:
    def __init__(self, unet_g, unet_u, guidance_scale: float, batch_total: int):
        self.g, self.u, self.guidance_scale, self.batch_total = unet_g, unet_u, guidance_scale, batch_total

    def __call__(self, latents: Tensor, t) -> Tensor:
        g = self.g(latents, t)
        u = self.u(latents, t)
        return u + self.guidance_scale * (g - u)


class UNetWithEmbeddings:
    """Binds encoder_hidden_states and takes `.sample` (reference unet/core.py:242-274)."""

    def __init__(self, unet, text_embeddings: Tensor, added_cond_kwargs: Optional[dict] = None):
        self.unet, self.text_embeddings, self.added = unet, text_embeddings, added_cond_kwargs

    def __call__(self, latents: Tensor, t) -> Tensor:
        if self.added is not None:   # SDXL "text_time" conditioning (pooled text embedding + size / crop ids)
            return self.unet(latents, t, encoder_hidden_states=self.text_embeddings, added_cond_kwargs=self.added).sample
        return self.unet(latents, t, encoder_hidden_states=self.text_embeddings).sample


class UnetWithExtraChannels:
    """Concatenates fixed extra channels (runway inpaint: mask + masked-image latents), reference
    unet/core.py:15-37; the extra channels are repeated when the CFG wrapper doubled the batch."""

    def __init__(self, unet, extra_channels: Tensor):
        self.unet, self.extra = unet, extra_channels

    def __call__(self, latents: Tensor, t) -> Tensor:
        extra = self.extra
        if extra.shape[0] != latents.shape[0]:
            extra = extra.repeat(latents.shape[0] // extra.shape[0], 1, 1, 1)
        return self.unet(torch.cat([latents, extra.to(latents.dtype)], dim=1), t)


# ------------------------------------------------------------------------------
# discrete sigma schedule (host tables, fp32)
# ------------------------------------------------------------------------------
class DiscreteSchedule:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.log_sigmas = self.sigmas.log()

    @property
    def sigma_min(self) -> Tensor:
        return self.sigmas[0]

    @property
    def sigma_max(self) -> Tensor:
        return self.sigmas[-1]

    def sigma_to_t(self, sigma: Tensor) -> Tensor:  # quantised: nearest training timestep in log-sigma
        log_sigma = sigma.log()
        dists = log_sigma - self.log_sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def t_to_sigma(self, t: Tensor) -> Tensor:
        t = t.float()
        low, high, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[low] + w * self.log_sigmas[high]).exp()


class KDiffusionUNetWrapper:
    """x0 = x + eps(x * c_in, t(sigma)) * c_out, c_in = (sigma^2+1)^-1/2, c_out = -sigma.
    `sigma` is a host fp32 scalar tensor (or [B] with equal entries, as the samplers pass)."""

    def __init__(self, eps_unet: Callable[[Tensor, object], Tensor], schedule: DiscreteSchedule):
        self.inner_model, self.schedule = eps_unet, schedule
        self.evals = 0

    def __call__(self, x: Tensor, sigma: Tensor) -> Tensor:
        s = sigma.reshape(-1)[0].to("cpu", torch.float32)
        c_in = float(1.0 / (s ** 2 + 1.0) ** 0.5)
        t = int(self.schedule.sigma_to_t(s))
        self.evals += 1
        eps = self.inner_model(x * c_in, t)
        return x + eps * (-float(s))


class KDiffusionVUNetWrapper(KDiffusionUNetWrapper):
    """v-prediction models (SD2.x 768-v): reference common_scheduler.py:350-355 over [3P] k_diffusion
    DiscreteVDDPMDenoiser with sigma_data = 1:  x0 = v(x * c_in, t(sigma)) * c_out + x * c_skip,
    c_skip = 1/(sigma^2+1), c_out = -sigma/sqrt(sigma^2+1), c_in = 1/sqrt(sigma^2+1)."""

    def __call__(self, x: Tensor, sigma: Tensor) -> Tensor:
        s = sigma.reshape(-1)[0].to("cpu", torch.float32)
        d = float(s ** 2 + 1.0)
        t = int(self.schedule.sigma_to_t(s))
        self.evals += 1
        v = self.inner_model(x * (1.0 / d ** 0.5), t)
        return v * (-float(s) / d ** 0.5) + x * (1.0 / d)


# ------------------------------------------------------------------------------
# samplers.  `model(x, sigma)` returns the denoised prediction; sigmas is a HOST fp32 tensor.
# ------------------------------------------------------------------------------
def _f(v) -> float:
    return float(v)


@torch.no_grad()
def sample_dpmpp_2m(model, x: Tensor, sigmas: Tensor, callback=None, warmup_lms: bool = False,
                    ddim_cutoff: float = 0.0, step_cb=None, **_):
    """DPM-Solver++(2M) with the reference's LMS warm-up (an extra model eval at the first step) and
    first-order cutoff: gyre/pipeline/schedulers/sample_dpmpp_2m.py:6-50."""
    sigmas = sigmas.to("cpu", torch.float32)
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    old_denoised = None
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        denoised = model(x, sigmas[i])
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        h = t_next - t
        if old_denoised is None and warmup_lms:
            r = 1 / 2
            s = t + r * h
            x_2 = _f(sigma_fn(s) / sigma_fn(t)) * x - _f((-h * r).expm1()) * denoised
            denoised_i = model(x_2, sigma_fn(s))
        elif sigmas[i + 1] <= ddim_cutoff or old_denoised is None:
            denoised_i = denoised
        else:
            h_last = t - t_fn(sigmas[i - 1])
            r = h_last / h
            denoised_i = _f(1 + 1 / (2 * r)) * denoised - _f(1 / (2 * r)) * old_denoised
        x = _f(sigma_fn(t_next) / sigma_fn(t)) * x - _f((-h).expm1()) * denoised_i
        old_denoised = denoised
    return x


def get_ancestral_step(sigma_from: Tensor, sigma_to: Tensor, eta: float = 1.0):
    if not eta:
        return sigma_to, torch.zeros(())
    sigma_up = torch.minimum(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def _churn(x: Tensor, sigmas: Tensor, i: int, s_churn: float, s_tmin: float, s_tmax: float, s_noise: float, noise_sampler):
    """Karras et al. (2022) Alg. 2 "churn" as in [3P] k_diffusion sample_euler / heun / dpm_2 (the reference passes
    s_churn / s_tmin / s_tmax from SchedulerConfig, common_scheduler.py:582-588): raise the noise level to
    sigma_hat = sigma * (1 + gamma) by adding fresh noise before the step.  The noise comes from the per-image
    generators (the reference routes torch.randn_like there through TorchRandOverride, randtools.py:67-141)."""
    sigma = sigmas[i]
    gamma = min(s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= float(sigma) <= s_tmax else 0.0
    sigma_hat = sigma * (gamma + 1)
    if gamma > 0:
        if noise_sampler is None:
            raise ValueError("churn needs a noise_sampler")
        x = x + noise_sampler(sigma, sigma_hat) * (s_noise * _f((sigma_hat ** 2 - sigma ** 2) ** 0.5))
    return x, sigma_hat


@torch.no_grad()
def sample_euler(model, x: Tensor, sigmas: Tensor, callback=None, step_cb=None, s_churn: float = 0.0, s_tmin: float = 0.0,
                 s_tmax: float = float("inf"), s_noise: float = 1.0, noise_sampler=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        x, sigma_hat = _churn(x, sigmas, i, s_churn, s_tmin, s_tmax, s_noise, noise_sampler)
        denoised = model(x, sigma_hat)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigma_hat, "denoised": denoised})
        d = (x - denoised) / _f(sigma_hat)
        x = x + d * _f(sigmas[i + 1] - sigma_hat)
    return x


@torch.no_grad()
def sample_euler_ancestral(model, x: Tensor, sigmas: Tensor, noise_sampler=None, callback=None, eta: float = 1.0,
                           s_noise: float = 1.0, step_cb=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        denoised = model(x, sigmas[i])
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        d = (x - denoised) / _f(sigmas[i])
        x = x + d * _f(sigma_down - sigmas[i])
        if sigmas[i + 1] > 0:
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * (s_noise * _f(sigma_up))
    return x


@torch.no_grad()
def sample_heun(model, x: Tensor, sigmas: Tensor, callback=None, step_cb=None, s_churn: float = 0.0, s_tmin: float = 0.0,
                s_tmax: float = float("inf"), s_noise: float = 1.0, noise_sampler=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        x, sigma_hat = _churn(x, sigmas, i, s_churn, s_tmin, s_tmax, s_noise, noise_sampler)
        denoised = model(x, sigma_hat)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigma_hat, "denoised": denoised})
        d = (x - denoised) / _f(sigma_hat)
        dt = _f(sigmas[i + 1] - sigma_hat)
        if sigmas[i + 1] == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            denoised_2 = model(x_2, sigmas[i + 1])
            d_2 = (x_2 - denoised_2) / _f(sigmas[i + 1])
            x = x + (d + d_2) / 2 * dt
    return x


@torch.no_grad()
def sample_dpm_2(model, x: Tensor, sigmas: Tensor, callback=None, step_cb=None, s_churn: float = 0.0, s_tmin: float = 0.0,
                 s_tmax: float = float("inf"), s_noise: float = 1.0, noise_sampler=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        x, sigma_hat = _churn(x, sigmas, i, s_churn, s_tmin, s_tmax, s_noise, noise_sampler)
        denoised = model(x, sigma_hat)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigma_hat, "denoised": denoised})
        d = (x - denoised) / _f(sigma_hat)
        if sigmas[i + 1] == 0:
            x = x + d * _f(sigmas[i + 1] - sigma_hat)
        else:
            sigma_mid = sigma_hat.log().lerp(sigmas[i + 1].log(), 0.5).exp()
            x_2 = x + d * _f(sigma_mid - sigma_hat)
            denoised_2 = model(x_2, sigma_mid)
            d_2 = (x_2 - denoised_2) / _f(sigma_mid)
            x = x + d_2 * _f(sigmas[i + 1] - sigma_hat)
    return x


@torch.no_grad()
def sample_dpmpp_2s_ancestral(model, x: Tensor, sigmas: Tensor, noise_sampler=None, callback=None, eta: float = 1.0,
                              s_noise: float = 1.0, step_cb=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        denoised = model(x, sigmas[i])
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        if sigma_down == 0:
            d = (x - denoised) / _f(sigmas[i])
            x = x + d * _f(sigma_down - sigmas[i])
        else:
            t, t_next = t_fn(sigmas[i]), t_fn(sigma_down)
            r = 1 / 2
            h = t_next - t
            s = t + r * h
            x_2 = _f(sigma_fn(s) / sigma_fn(t)) * x - _f((-h * r).expm1()) * denoised
            denoised_2 = model(x_2, sigma_fn(s))
            x = _f(sigma_fn(t_next) / sigma_fn(t)) * x - _f((-h).expm1()) * denoised_2
        if sigmas[i + 1] > 0:
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * (s_noise * _f(sigma_up))
    return x


SAMPLERS = {
    # name -> (function, kwargs bound like gyre/pipeline/samplers.py:47-67)
    "euler": (sample_euler, {}),
    "euler_a": (sample_euler_ancestral, {}),
    "heun": (sample_heun, {}),
    "dpm_2": (sample_dpm_2, {}),
    "dpmpp_2s_a": (sample_dpmpp_2s_ancestral, {}),
    "dpmpp_2m": (sample_dpmpp_2m, {"warmup_lms": True, "ddim_cutoff": 0.1}),
}


def get_sigmas_karras(n: int, sigma_min: float, sigma_max: float, rho: float = 7.0) -> Tensor:
    ramp = torch.linspace(0, 1, n)
    min_inv, max_inv = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return torch.cat([(max_inv + ramp * (min_inv - max_inv)) ** rho, torch.zeros(1)])


@dataclass
class SchedulerConfig:
    eta: Optional[float] = None
    churn: Optional[float] = None
    churn_tmin: float = 0.0
    churn_tmax: float = float("inf")
    sigma_min: Optional[float] = None
    sigma_max: Optional[float] = None
    karras_rho: Optional[float] = None
    noise_type: str = "normal"                                   # "normal" | "brownian" (common_scheduler.py:26,94)


class KDiffusionScheduler:
    """The loop driver: schedule, img2img start offset, initial noise scaling, sampler call."""

    def __init__(self, sampler: str | Callable, generators: List[torch.Generator], device, dtype=torch.float32):
        if callable(sampler):
            self.sampler_fn, self.sampler_kwargs = sampler, {}
        else:
            if sampler not in SAMPLERS:
                raise NotImplementedError(f"sampler {sampler!r} is not implemented (have {sorted(SAMPLERS)})")
            self.sampler_fn, self.sampler_kwargs = SAMPLERS[sampler]
        self.generators, self.device, self.dtype = generators, device, dtype
        self.schedule = DiscreteSchedule()
        self.eps_unet = None
        self.eps_unets = []
        self.unets: List[KDiffusionUNetWrapper] = []
        self.unet: Optional[KDiffusionUNetWrapper] = None
        self.sigmas: Optional[Tensor] = None
        self.start_offset = 0
        self.eta = None

    def set_eps_unet(self, eps_unet):
        self.set_eps_unets([eps_unet])

    def set_eps_unets(self, eps_unets):
        """One denoiser per mode-tree leaf (reference common_scheduler.py set_eps_unets, unified_pipeline.py:2436):
        the hires fix runs a natural-size and a full-size leaf through the same schedule."""
        self.eps_unets = list(eps_unets)
        self.eps_unet = self.eps_unets[-1] if self.eps_unets else None

    def set_timesteps(self, num_inference_steps: int, start_offset: Optional[int] = None,
                      strength: Optional[float] = None, config: SchedulerConfig = SchedulerConfig(),
                      prediction_type: str = "epsilon"):
        if self.eps_unet is None:
            raise ValueError("Epsilon unet needs to be set before timesteps")
        if prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError(f"prediction_type {prediction_type!r}")
        s = self.schedule
        wrapper = KDiffusionVUNetWrapper if prediction_type == "v_prediction" else KDiffusionUNetWrapper
        self.unets = [wrapper(e, s) for e in self.eps_unets]
        self.unet = self.unets[-1]
        sigma_min, sigma_max = config.sigma_min, config.sigma_max
        if sigma_min is not None:
            sigma_min = max(float(s.sigma_min), sigma_min)
        if sigma_max is not None:
            sigma_max = min(float(s.sigma_max), sigma_max)
        if config.karras_rho is not None:
            q = lambda v: float(s.t_to_sigma(s.sigma_to_t(torch.tensor(v))))
            self.sigmas = get_sigmas_karras(num_inference_steps,
                                            q(sigma_min) if sigma_min is not None else float(s.sigma_min),
                                            q(sigma_max) if sigma_max is not None else float(s.sigma_max),
                                            config.karras_rho)
        else:
            t_min = 0 if sigma_min is None else float(s.sigma_to_t(torch.tensor(sigma_min)))
            t_max = len(s.sigmas) - 1 if sigma_max is None else float(s.sigma_to_t(torch.tensor(sigma_max)))
            t = torch.linspace(t_max, t_min, num_inference_steps)
            self.sigmas = torch.cat([s.t_to_sigma(t), torch.zeros(1)])
        self.eta = config.eta
        if config.noise_type not in ("normal", "brownian"):
            raise ValueError(f"noise_type {config.noise_type!r}")
        self.noise_type = config.noise_type
        self.churn, self.churn_tmin, self.churn_tmax = config.churn, config.churn_tmin, config.churn_tmax
        self.num_inference_steps = num_inference_steps
        if strength is not None:
            if start_offset is not None:
                raise ValueError("Can't pass both start_offset and strength to set_timesteps")
            init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
            self.start_offset = max(num_inference_steps - init_timestep, 0)
        else:
            self.start_offset = start_offset or 0
        self.start_sigma = self.sigmas[self.start_offset]

    def prepare_initial_latents(self, latents: Tensor) -> Tensor:
        return latents * float(self.sigmas[0])

    def add_noise(self, latents: Tensor, noise: Tensor) -> Tensor:
        # reference adds sigma(t(start sigma)) (quantised), common_scheduler.py:550-553 + :538
        s = self.schedule
        sigma = s.t_to_sigma(s.sigma_to_t(self.start_sigma))
        return latents + noise * float(sigma)

    def loop(self, latents: Tensor, callback=None, k_wrap=None, k_model=None) -> Tensor:
        """k_wrap(px0, u) -> px0 lets a mode post-process the denoised prediction with the progress value u in
        [0, 0.999] (reference KDiffusionPositionTracker, common_scheduler.py:358-389, and Mode.wrap_k_unet).
        k_model(x, sigma, u) -> px0 replaces the whole denoiser (collapsed mode tree: hires fix / graft over
        self.unets)."""
        sigmas = self.sigmas[self.start_offset:]
        kwargs = dict(self.sampler_kwargs)
        by_range, wants_n = kwargs.pop("_range", False), kwargs.pop("_n", False)
        if self.eta is not None:
            kwargs["eta"] = self.eta
        if getattr(self, "churn", None):
            kwargs.update(s_churn=self.churn, s_tmin=self.churn_tmin, s_tmax=self.churn_tmax)
        if getattr(self, "noise_type", "normal") == "brownian":
            # common_scheduler.py:563-564,597-606: one seed per image from that image's generator; the tree spans the sigmas
            # of THIS loop (after the img2img start offset)
            seeds = [int(torch.randint(0, 2 ** 63 - 1, [], generator=g, device=g.device).item()) for g in self.generators]
            kwargs["noise_sampler"] = BrownianTreeNoiseSampler(latents, sigmas[sigmas > 0].min(), sigmas.max(), seed=seeds)
        else:
            kwargs["noise_sampler"] = lambda _, __: batched_randn(latents.shape, self.generators, self.device, self.dtype)
        model = self.unet
        if k_wrap is not None or k_model is not None:
            u_off = self.start_offset / len(self.sigmas)
            state = {"i": None, "i_max": len(sigmas) - 1}
            host_sigmas = sigmas.to("cpu", torch.float32)

            def tracked(x, sigma):
                i = state["i"]
                if i is None:     # sampler without a fixed step range (dpm_fast / dpm_adaptive): place sigma in the
                    i = int((host_sigmas >= float(sigma.reshape(-1)[0])).sum())      # schedule (common_scheduler.py:366-377)
                u = u_off + (1 - u_off) * i / state["i_max"]
                u = max(min(u, 0.999), 0)
                if k_model is not None:
                    return k_model(x, sigma, u)
                return k_wrap(self.unet(x, sigma), u)

            model = tracked
            kwargs["step_cb"] = lambda i: state.__setitem__("i", i)
        if by_range:
            if wants_n:
                kwargs["n"] = self.num_inference_steps
            pos = sigmas[sigmas > 0]
            return self.sampler_fn(model, latents, pos.min(), sigmas.max(), callback=callback, **kwargs)
        return self.sampler_fn(model, latents, sigmas, callback=callback, **kwargs)


# ------------------------------------------------------------------------------
# more k-diffusion samplers [3P k_diffusion.sampling], selected at reference samplers.py:48-57
# ------------------------------------------------------------------------------
@torch.no_grad()
def sample_dpm_2_ancestral(model, x: Tensor, sigmas: Tensor, noise_sampler=None, callback=None, eta: float = 1.0,
                           s_noise: float = 1.0, step_cb=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        denoised = model(x, sigmas[i])
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        d = (x - denoised) / _f(sigmas[i])
        if sigma_down == 0:
            x = x + d * _f(sigma_down - sigmas[i])
        else:
            sigma_mid = sigmas[i].log().lerp(sigma_down.log(), 0.5).exp()
            x_2 = x + d * _f(sigma_mid - sigmas[i])
            denoised_2 = model(x_2, sigma_mid)
            d_2 = (x_2 - denoised_2) / _f(sigma_mid)
            x = x + d_2 * _f(sigma_down - sigmas[i])
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * (s_noise * _f(sigma_up))
    return x


def linear_multistep_coeff(order: int, t, i: int, j: int) -> float:
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")

    def fn(tau):
        prod = 1.0
        for k in range(order):
            if j == k:
                continue
            prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return prod

    return integrate.quad(fn, t[i], t[i + 1], epsrel=1e-4)[0]


@torch.no_grad()
def sample_lms(model, x: Tensor, sigmas: Tensor, callback=None, order: int = 4, step_cb=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    sig = sigmas.double().numpy()
    ds = []
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        denoised = model(x, sigmas[i])
        d = (x - denoised) / _f(sigmas[i])
        ds.append(d)
        if len(ds) > order:
            ds.pop(0)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        cur_order = min(i + 1, order)
        coeffs = [linear_multistep_coeff(cur_order, sig, i, j) for j in range(cur_order)]
        x = x + sum(c * dd for c, dd in zip(coeffs, reversed(ds)))
    return x


@torch.no_grad()
def sample_dpmpp_sde(model, x: Tensor, sigmas: Tensor, noise_sampler=None, callback=None, eta: float = 1.0,
                     s_noise: float = 1.0, r: float = 0.5, step_cb=None, **_):
    """DPM-Solver++ (stochastic), [3P k_diffusion.sampling.sample_dpmpp_sde] selected at reference samplers.py:56.
    The reference's default noise source is its per-image batched_randn (common_scheduler.py:608-610); the Brownian
    tree (torchsde) variant is not available."""
    sigmas = sigmas.to("cpu", torch.float32)
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        denoised = model(x, sigmas[i])
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        if sigmas[i + 1] == 0:
            d = (x - denoised) / _f(sigmas[i])
            x = x + d * _f(sigmas[i + 1] - sigmas[i])
            continue
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        h = t_next - t
        s = t + h * r
        fac = 1 / (2 * r)
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(s), eta)
        s_ = t_fn(sd)
        x_2 = _f(sigma_fn(s_) / sigma_fn(t)) * x - _f((t - s_).expm1()) * denoised
        x_2 = x_2 + noise_sampler(sigma_fn(t), sigma_fn(s)) * (s_noise * _f(su))
        denoised_2 = model(x_2, sigma_fn(s))
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(t_next), eta)
        t_next_ = t_fn(sd)
        denoised_d = (1 - fac) * denoised + fac * denoised_2
        x = _f(sigma_fn(t_next_) / sigma_fn(t)) * x - _f((t - t_next_).expm1()) * denoised_d
        x = x + noise_sampler(sigma_fn(t), sigma_fn(t_next)) * (s_noise * _f(su))
    return x


class DPMSolver:
    """DPM-Solver (Lu et al. 2022) in k-diffusion's sigma parameterisation t = -log sigma, with its single-step
    orders 1-3, the fixed-budget schedule (dpm_fast) and the PID step-size controller (dpm_adaptive)
    [3P k_diffusion.sampling.DPMSolver; selected at reference samplers.py:54-55].  t values are host fp32 scalars."""

    def __init__(self, model, info_callback=None):
        self.model, self.info_callback = model, info_callback

    @staticmethod
    def t(sigma: Tensor) -> Tensor:
        return sigma.log().neg()

    @staticmethod
    def sigma(t: Tensor) -> Tensor:
        return t.neg().exp()

    def eps(self, cache: dict, key: str, x: Tensor, t: Tensor):
        if key in cache:
            return cache[key], cache
        sig = self.sigma(t)
        e = (x - self.model(x, sig)) / _f(sig)
        return e, {key: e, **cache}

    def step1(self, x, t, t_next, cache=None):
        cache = {} if cache is None else cache
        h = t_next - t
        e, cache = self.eps(cache, "eps", x, t)
        return x - _f(self.sigma(t_next) * h.expm1()) * e, cache

    def step2(self, x, t, t_next, r1=1 / 2, cache=None):
        cache = {} if cache is None else cache
        h = t_next - t
        e, cache = self.eps(cache, "eps", x, t)
        s1 = t + r1 * h
        u1 = x - _f(self.sigma(s1) * (r1 * h).expm1()) * e
        e1, cache = self.eps(cache, "eps_r1", u1, s1)
        x2 = x - _f(self.sigma(t_next) * h.expm1()) * e - _f(self.sigma(t_next) / (2 * r1) * h.expm1()) * (e1 - e)
        return x2, cache

    def step3(self, x, t, t_next, r1=1 / 3, r2=2 / 3, cache=None):
        cache = {} if cache is None else cache
        h = t_next - t
        e, cache = self.eps(cache, "eps", x, t)
        s1, s2 = t + r1 * h, t + r2 * h
        u1 = x - _f(self.sigma(s1) * (r1 * h).expm1()) * e
        e1, cache = self.eps(cache, "eps_r1", u1, s1)
        u2 = x - _f(self.sigma(s2) * (r2 * h).expm1()) * e \
            - _f(self.sigma(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1)) * (e1 - e)
        e2, cache = self.eps(cache, "eps_r2", u2, s2)
        x3 = x - _f(self.sigma(t_next) * h.expm1()) * e - _f(self.sigma(t_next) / r2 * (h.expm1() / h - 1)) * (e2 - e)
        return x3, cache

    def _ancestral(self, t, t_next, t_end, eta):
        if not eta:
            return t_next, 0.0
        sd, _ = get_ancestral_step(self.sigma(t), self.sigma(t_next), eta)
        t_ = torch.minimum(t_end, self.t(sd))
        su = (self.sigma(t_next) ** 2 - self.sigma(t_) ** 2) ** 0.5
        return t_, _f(su)

    def fast(self, x, t_start, t_end, nfe: int, eta=0.0, s_noise=1.0, noise_sampler=None):
        if not t_end > t_start and eta:
            raise ValueError("eta must be 0 for reverse sampling")
        m = nfe // 3 + 1
        ts = torch.linspace(float(t_start), float(t_end), m + 1)
        orders = [3] * (m - 2) + [2, 1] if nfe % 3 == 0 else [3] * (m - 1) + [nfe % 3]
        for i, order in enumerate(orders):
            t, t_next = ts[i], ts[i + 1]
            t_next_, su = self._ancestral(t, t_next, t_end, eta)
            e, cache = self.eps({}, "eps", x, t)
            if self.info_callback is not None:
                self.info_callback({"x": x, "i": i, "t": t, "t_up": t, "denoised": x - _f(self.sigma(t)) * e})
            x, _ = (self.step1, self.step2, self.step3)[order - 1](x, t, t_next_, cache=cache)
            if su:
                x = x + noise_sampler(self.sigma(t), self.sigma(t_next)) * (su * s_noise)
        return x

    def adaptive(self, x, t_start, t_end, order=3, rtol=0.05, atol=0.0078, h_init=0.05, pcoeff=0.0, icoeff=1.0,
                 dcoeff=0.0, accept_safety=0.81, eta=0.0, s_noise=1.0, noise_sampler=None):
        import math
        if order not in (2, 3):
            raise ValueError("order should be 2 or 3")
        forward = bool(t_end > t_start)
        if not forward and eta:
            raise ValueError("eta must be 0 for reverse sampling")
        h = abs(h_init) * (1 if forward else -1)
        ctl_order = 1.5 if eta else order
        b1, b2, b3 = (pcoeff + icoeff + dcoeff) / ctl_order, -(pcoeff + 2 * dcoeff) / ctl_order, dcoeff / ctl_order
        errs = []
        s, x_prev = t_start, x
        info = {"steps": 0, "nfe": 0, "n_accept": 0, "n_reject": 0}
        while (s < t_end - 1e-5) if forward else (s > t_end + 1e-5):
            t = torch.minimum(t_end, s + h) if forward else torch.maximum(t_end, s + h)
            t_, su = self._ancestral(s, t, t_end, eta)
            e, cache = self.eps({}, "eps", x, s)
            if order == 2:
                x_low, cache = self.step1(x, s, t_, cache=cache)
                x_high, cache = self.step2(x, s, t_, cache=cache)
            else:
                x_low, cache = self.step2(x, s, t_, r1=1 / 3, cache=cache)
                x_high, cache = self.step3(x, s, t_, cache=cache)
            delta = torch.maximum(torch.tensor(atol, device=x.device, dtype=x.dtype),
                                  rtol * torch.maximum(x_low.abs(), x_prev.abs()))
            error = float(torch.linalg.norm((x_low - x_high) / delta)) / x.numel() ** 0.5
            inv = 1 / (error + 1e-8)                                     # PID step-size controller
            if not errs:
                errs = [inv, inv, inv]
            errs[0] = inv
            factor = 1 + math.atan(errs[0] ** b1 * errs[1] ** b2 * errs[2] ** b3 - 1)
            accept = factor >= accept_safety
            if accept:
                errs[2], errs[1] = errs[1], errs[0]
                x_prev = x_low
                x = x_high
                if su:
                    x = x + noise_sampler(self.sigma(s), self.sigma(t)) * (su * s_noise)
                s = t
                info["n_accept"] += 1
            else:
                info["n_reject"] += 1
            h *= factor
            info["nfe"] += order
            info["steps"] += 1
            if self.info_callback is not None:
                self.info_callback({"x": x, "i": info["steps"] - 1, "t": s, "t_up": s, "error": error, "h": h, **info})
        return x, info


def _solver_callback(solver: "DPMSolver", callback):
    if callback is None:
        return None
    return lambda info: callback({"sigma": solver.sigma(info["t"]), "sigma_hat": solver.sigma(info["t_up"]), **info})


@torch.no_grad()
def sample_dpm_fast(model, x: Tensor, sigma_min, sigma_max, n: int, callback=None, eta: float = 0.0,
                    s_noise: float = 1.0, noise_sampler=None, **_):
    """DPM-Solver-Fast: fixed budget of n model evaluations between sigma_max and sigma_min."""
    sigma_min, sigma_max = torch.as_tensor(sigma_min, dtype=torch.float32).cpu(), torch.as_tensor(sigma_max, dtype=torch.float32).cpu()
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    solver = DPMSolver(model)
    solver.info_callback = _solver_callback(solver, callback)
    return solver.fast(x, solver.t(sigma_max), solver.t(sigma_min), n, eta, s_noise, noise_sampler)


@torch.no_grad()
def sample_dpm_adaptive(model, x: Tensor, sigma_min, sigma_max, callback=None, order: int = 3, rtol: float = 0.05,
                        atol: float = 0.0078, h_init: float = 0.05, pcoeff: float = 0.0, icoeff: float = 1.0,
                        dcoeff: float = 0.0, accept_safety: float = 0.81, eta: float = 0.0, s_noise: float = 1.0,
                        noise_sampler=None, return_info: bool = False, **_):
    """DPM-Solver-12 / -23 with adaptive step size (the number of model evaluations is data dependent; the error norm
    is taken over the whole batch, so - as in the reference - this sampler is not batch independent)."""
    sigma_min, sigma_max = torch.as_tensor(sigma_min, dtype=torch.float32).cpu(), torch.as_tensor(sigma_max, dtype=torch.float32).cpu()
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    solver = DPMSolver(model)
    solver.info_callback = _solver_callback(solver, callback)
    x, info = solver.adaptive(x, solver.t(sigma_max), solver.t(sigma_min), order, rtol, atol, h_init, pcoeff, icoeff,
                              dcoeff, accept_safety, eta, s_noise, noise_sampler)
    return (x, info) if return_info else x


SAMPLERS.update({
    "dpm_2_a": (sample_dpm_2_ancestral, {}),
    "lms": (sample_lms, {}),
    "dpmpp_sde": (sample_dpmpp_sde, {}),
    # these two take (sigma_min, sigma_max[, n]) instead of a sigma schedule (common_scheduler.py:590-594)
    "dpm_fast": (sample_dpm_fast, {"_range": True, "_n": True}),
    "dpm_adaptive": (sample_dpm_adaptive, {"_range": True}),
})


# ------------------------------------------------------------------------------
# diffusers-scheduler loop (reference common_scheduler.py:179-314: DDIM, PLMS = PNDM(skip_prk_steps=True),
# DPMSolverMultistep orders 1-3).
# The step arithmetic is diffusers' [3P, ~=0.16.0, not vendored]; restated from the published algorithms with the
# SD1.x scheduler_config.json values (scaled_linear betas, steps_offset=1, set_alpha_to_one=False, no clipping).
# Coefficients are host fp32/64 scalars applied to the device latents (no per-step sync).
# ------------------------------------------------------------------------------
DIFFUSERS_SAMPLERS = ("ddim", "plms", "dpmsolverpp_1", "dpmsolverpp_2", "dpmsolverpp_3")


class DiffusersLikeScheduler:
    init_noise_sigma = 1.0

    def __init__(self, kind: str, num_train_timesteps: int = 1000, steps_offset: int = 1):
        if kind not in DIFFUSERS_SAMPLERS:
            raise NotImplementedError(kind)
        self.kind, self.T, self.steps_offset = kind, num_train_timesteps, steps_offset
        self.solver_order = int(kind[-1]) if kind.startswith("dpmsolverpp_") else 0
        self.alphas_cumprod = DiscreteSchedule(num_train_timesteps).alphas_cumprod.double()
        self.final_alpha_cumprod = self.alphas_cumprod[0]  # set_alpha_to_one = False

    def set_timesteps(self, n: int):
        import numpy as np
        self.n = n
        if self.solver_order:
            # DPMSolverMultistepScheduler (diffusers ~0.16): n+1 points over [0, T-1], first one dropped at the end
            self.timesteps = np.linspace(0, self.T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
            self.outputs, self.lower_order_nums = [], 0
            return
        ratio = self.T // n
        base = (np.arange(0, n) * ratio).round().astype(np.int64) + self.steps_offset
        if self.kind == "ddim":
            self.timesteps = base[::-1].copy()
        else:  # PNDM with skip_prk_steps: the second timestep is visited twice
            self.timesteps = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.ets, self.counter, self.cur_sample = [], 0, None

    def _ac(self, t: int) -> float:
        return float(self.alphas_cumprod[t]) if t >= 0 else float(self.final_alpha_cumprod)

    # -- DPM-Solver++ multistep (Lu et al. 2022; reference samplers.py:34-45 selects solver_order 1/2/3 with the
    # diffusers defaults: algorithm "dpmsolver++", midpoint solver, lower_order_final) -------------------------
    def _lambda_alpha_sigma(self, t: int):
        import math
        a = float(self.alphas_cumprod[t])
        alpha, sigma = math.sqrt(a), math.sqrt(1 - a)
        return math.log(alpha) - math.log(sigma), alpha, sigma

    def _dpmsolver_step(self, eps: Tensor, t: int, sample: Tensor) -> Tensor:
        import math
        ts = [int(v) for v in self.timesteps]
        idx = ts.index(int(t))
        last = idx == len(ts) - 1
        prev_t = 0 if last else ts[idx + 1]
        small = len(ts) < 15
        lower_final = last and small
        lower_second = idx == len(ts) - 2 and small
        lam_s0, alpha_s0, sigma_s0 = self._lambda_alpha_sigma(t)
        x0 = (sample - sigma_s0 * eps) / alpha_s0                       # data prediction from the eps model
        # history = data predictions only; their timesteps are read off the SCHEDULE (timesteps[idx - 1], [idx - 2]) as the
        # pinned DPMSolverMultistepScheduler.step does - identical for a single UNet, and what keeps two grafted leaves
        # that step this shared object at the same timestep finite (reference common_scheduler.py:240,261-283)
        self.outputs = (self.outputs + [x0])[-max(self.solver_order, 1):]
        lam_t, alpha_t, sigma_t = self._lambda_alpha_sigma(prev_t)
        h = lam_t - lam_s0
        em1 = math.expm1(-h)                                            # e^{-h} - 1
        order = self.solver_order
        if order == 1 or self.lower_order_nums < 1 or lower_final:
            use = 1
        elif order == 2 or self.lower_order_nums < 2 or lower_second:
            use = 2
        else:
            use = 3
        # the history's timesteps are read off the schedule: a leaf that starts stepping while another one has already raised
        # lower_order_nums would read timesteps[-1] / [-2] (the END of the schedule) for its first steps
        use = min(use, idx + 1, len(self.outputs))
        m0 = self.outputs[-1]
        out = (sigma_t / sigma_s0) * sample - (alpha_t * em1) * m0
        if use >= 2:
            s1, m1 = ts[idx - 1], self.outputs[-2]
            lam_s1 = self._lambda_alpha_sigma(s1)[0]
            r0 = (lam_s0 - lam_s1) / h
            d1_0 = (m0 - m1) * (1.0 / r0)
            if use == 2:
                out = out - (0.5 * alpha_t * em1) * d1_0
            else:
                s2, m2 = ts[idx - 2], self.outputs[-3]
                lam_s2 = self._lambda_alpha_sigma(s2)[0]
                r1 = (lam_s1 - lam_s2) / h
                d1_1 = (m1 - m2) * (1.0 / r1)
                d1 = d1_0 + (d1_0 - d1_1) * (r0 / (r0 + r1))
                d2 = (d1_0 - d1_1) * (1.0 / (r0 + r1))
                out = out + (alpha_t * (em1 / h + 1.0)) * d1 - (alpha_t * ((em1 + h) / (h * h) - 0.5)) * d2
        if self.lower_order_nums < order:
            self.lower_order_nums += 1
        return out

    def _check_history(self, history, sample: Tensor) -> None:
        """PLMS / DPM-Solver++ keep ONE multistep history per scheduler object.  Leaves of a mode tree that share the object
        (reference common_scheduler.py:240) may feed it together only when their latents have one shape (grafted inpaint);
        a hires-fix tree steps a natural-size and a full-size leaf through it, and mixing their predictions cannot work -
        the reference fails there with a shape error from deep inside the step; say what is wrong instead."""
        for h in history:
            if h is not None and tuple(h.shape) != tuple(sample.shape):
                raise ValueError(f"sampler {self.kind!r} keeps a multistep history and cannot step mode-tree leaves of different "
                                 f"latent sizes ({tuple(h.shape)} vs {tuple(sample.shape)}: hires fix) through one scheduler; use "
                                 f"ddim or a k-diffusion sampler for this request")

    def step(self, eps: Tensor, t: int, sample: Tensor) -> Tensor:
        if self.solver_order:
            self._check_history(self.outputs, sample)
            return self._dpmsolver_step(eps, t, sample)
        if self.kind != "ddim":
            self._check_history(list(self.ets) + [self.cur_sample], sample)
        ratio = self.T // self.n
        if self.kind == "ddim":
            prev_t = t - ratio
            a_t, a_prev = self._ac(t), self._ac(prev_t)
            x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
            return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps  # eta = 0
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev_t = t
            t = t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            eps = (eps + self.ets[-1]) / 2
            sample, self.cur_sample = self.cur_sample, None
        elif len(self.ets) == 2:
            eps = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            eps = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            eps = (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4]) / 24
        self.counter += 1
        a_t, a_prev = self._ac(t), self._ac(prev_t)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return sample_coeff * sample - (a_prev - a_t) / denom * eps


class DiffusersScheduler:
    """Loop driver with the KDiffusionScheduler surface (set_eps_unet / set_timesteps / prepare_initial_latents /
    add_noise / loop) for the samplers k-diffusion has no twin for."""

    def __init__(self, kind: str, generators, device, dtype=torch.float32):
        self.sched = DiffusersLikeScheduler(kind)
        self.generators, self.device, self.dtype = generators, device, dtype
        self.eps_unet, self.eps_unets, self.unets, self.unet = None, [], [], None
        self.start_offset = 0

    def set_eps_unet(self, eps_unet):
        self.set_eps_unets([eps_unet])

    def set_eps_unets(self, eps_unets):
        """One noise predictor per mode-tree leaf (common_scheduler.py set_eps_unets).  Every leaf gets its own step
        wrapper over the ONE scheduler object, exactly as the reference builds them (common_scheduler.py:240
        ``self.unets = [self.wrap_unet(eps_unet) for eps_unet in self.eps_unets]``): DDIM is stateless, so a grafted
        inpaint runs as it should; PLMS / DPM-Solver++ keep a multistep history in that shared object, which the two
        leaves of a graft both feed while they overlap (0.1 < u < 0.3) - the reference's behaviour, reproduced."""
        self.eps_unets = list(eps_unets)
        self.eps_unet = self.eps_unets[-1] if self.eps_unets else None

    def set_timesteps(self, num_inference_steps: int, start_offset=None, strength=None, config=None,
                      prediction_type: str = "epsilon"):
        if self.eps_unet is None:
            raise ValueError("Epsilon unet needs to be set before timesteps")
        if prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError(f"prediction_type {prediction_type!r}")
        self.prediction_type = prediction_type
        self.sched.set_timesteps(num_inference_steps)
        if strength is not None:
            # common_scheduler.py:226-233: the scheduler config's steps_offset (1 for the SD DDIM / PNDM configs; the
            # pinned DPMSolverMultistepScheduler has no such field -> 0) enters twice, which only shows once
            # int(n * strength) reaches n: strength 1.0 then starts at timesteps[1], not timesteps[0]
            offset = self.sched.steps_offset if self.sched.solver_order == 0 else 0
            init = min(int(num_inference_steps * strength) + offset, num_inference_steps)
            self.start_offset = max(num_inference_steps - init + offset, 0)
        else:
            self.start_offset = start_offset or 0
        outer = self

        class _DUnet:
            """wrap_unet (common_scheduler.py:261-283): x_t, t -> eps -> scheduler.step -> x_{t-1}"""
            def __init__(self, eps_unet):
                self.eps_unet, self.evals = eps_unet, 0

            def __call__(self, x, t):
                t = int(t)
                eps = self.eps_unet(x, t)
                if outer.prediction_type == "v_prediction":
                    # eps = sqrt(abar) v + sqrt(1 - abar) x   (x0 = sqrt(abar) x - sqrt(1 - abar) v)
                    a = float(outer.sched.alphas_cumprod[t])
                    eps = a ** 0.5 * eps + (1 - a) ** 0.5 * x
                self.evals += 1
                return outer.sched.step(eps, t, x)

        self.unets = [_DUnet(e) for e in self.eps_unets]
        self.unet = self.unets[-1]

    def prepare_initial_latents(self, latents: Tensor) -> Tensor:
        return latents * self.sched.init_noise_sigma

    def add_noise(self, latents: Tensor, noise: Tensor) -> Tensor:
        a = float(self.sched.alphas_cumprod[int(self.sched.timesteps[self.start_offset])])
        return a ** 0.5 * latents + (1 - a) ** 0.5 * noise

    def add_noise_at(self, latents: Tensor, noise: Tensor, t: int) -> Tensor:
        a = float(self.sched.alphas_cumprod[int(t)])
        return a ** 0.5 * latents + (1 - a) ** 0.5 * noise

    def loop(self, latents: Tensor, callback=None, d_model=None) -> Tensor:
        """d_model(x_t, t, u) -> x_{t-1} replaces the plain step (a mode's wrap_d_unet around self.unets[0]); u is the
        progress value in [0, 0.999] (common_scheduler.py:285-301)."""
        x = latents
        ts = self.sched.timesteps[self.start_offset:]
        u_off = self.start_offset / max(len(self.sched.timesteps), 1)
        for i, t in enumerate(ts):
            if d_model is not None:
                u = u_off + (1 - u_off) * i / max(len(ts), 1)
                x = d_model(x, int(t), max(min(u, 0.999), 0))
            else:
                x = self.unet(x, int(t))
            if callback is not None:
                callback({"x": x, "i": i, "t": int(t)})
        return x


def make_scheduler(sampler, generators, device, dtype=torch.float32):
    """sampler name -> loop driver, the split the reference makes in build_sampler_set (samplers.py:110-130)."""
    if isinstance(sampler, str) and sampler in DIFFUSERS_SAMPLERS:
        return DiffusersScheduler(sampler, generators, device, dtype)
    return KDiffusionScheduler(sampler, generators, device, dtype)
