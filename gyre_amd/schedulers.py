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

    def __init__(self, unet, text_embeddings: Tensor):
        self.unet, self.text_embeddings = unet, text_embeddings

    def __call__(self, latents: Tensor, t) -> Tensor:
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


# ------------------------------------------------------------------------------
# samplers.  `model(x, sigma)` returns the denoised prediction; sigmas is a HOST fp32 tensor.
# ------------------------------------------------------------------------------
def _f(v) -> float:
    return float(v)


@torch.no_grad()
def sample_dpmpp_2m(model, x: Tensor, sigmas: Tensor, callback=None, warmup_lms: bool = False,
                    ddim_cutoff: float = 0.0, **_):
    """DPM-Solver++(2M) with the reference's LMS warm-up (an extra model eval at the first step) and
    first-order cutoff: gyre/pipeline/schedulers/sample_dpmpp_2m.py:6-50."""
    sigmas = sigmas.to("cpu", torch.float32)
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    old_denoised = None
    for i in range(len(sigmas) - 1):
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


@torch.no_grad()
def sample_euler(model, x: Tensor, sigmas: Tensor, callback=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i])
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        d = (x - denoised) / _f(sigmas[i])
        x = x + d * _f(sigmas[i + 1] - sigmas[i])
    return x


@torch.no_grad()
def sample_euler_ancestral(model, x: Tensor, sigmas: Tensor, noise_sampler=None, callback=None, eta: float = 1.0,
                           s_noise: float = 1.0, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
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
def sample_heun(model, x: Tensor, sigmas: Tensor, callback=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i])
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        d = (x - denoised) / _f(sigmas[i])
        dt = _f(sigmas[i + 1] - sigmas[i])
        if sigmas[i + 1] == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            denoised_2 = model(x_2, sigmas[i + 1])
            d_2 = (x_2 - denoised_2) / _f(sigmas[i + 1])
            x = x + (d + d_2) / 2 * dt
    return x


@torch.no_grad()
def sample_dpm_2(model, x: Tensor, sigmas: Tensor, callback=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i])
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        d = (x - denoised) / _f(sigmas[i])
        if sigmas[i + 1] == 0:
            x = x + d * _f(sigmas[i + 1] - sigmas[i])
        else:
            sigma_mid = sigmas[i].log().lerp(sigmas[i + 1].log(), 0.5).exp()
            x_2 = x + d * _f(sigma_mid - sigmas[i])
            denoised_2 = model(x_2, sigma_mid)
            d_2 = (x_2 - denoised_2) / _f(sigma_mid)
            x = x + d_2 * _f(sigmas[i + 1] - sigmas[i])
    return x


@torch.no_grad()
def sample_dpmpp_2s_ancestral(model, x: Tensor, sigmas: Tensor, noise_sampler=None, callback=None, eta: float = 1.0,
                              s_noise: float = 1.0, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    for i in range(len(sigmas) - 1):
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
    sigma_min: Optional[float] = None
    sigma_max: Optional[float] = None
    karras_rho: Optional[float] = None


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
        self.unet: Optional[KDiffusionUNetWrapper] = None
        self.sigmas: Optional[Tensor] = None
        self.start_offset = 0
        self.eta = None

    def set_eps_unet(self, eps_unet):
        self.eps_unet = eps_unet

    def set_timesteps(self, num_inference_steps: int, start_offset: Optional[int] = None,
                      strength: Optional[float] = None, config: SchedulerConfig = SchedulerConfig()):
        if self.eps_unet is None:
            raise ValueError("Epsilon unet needs to be set before timesteps")
        s = self.schedule
        self.unet = KDiffusionUNetWrapper(self.eps_unet, s)
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

    def loop(self, latents: Tensor, callback=None) -> Tensor:
        sigmas = self.sigmas[self.start_offset:]
        kwargs = dict(self.sampler_kwargs)
        if self.eta is not None:
            kwargs["eta"] = self.eta
        kwargs["noise_sampler"] = lambda _, __: batched_randn(latents.shape, self.generators, self.device, self.dtype)
        return self.sampler_fn(self.unet, latents, sigmas, callback=callback, **kwargs)
