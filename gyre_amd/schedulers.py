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


@torch.no_grad()
def sample_euler(model, x: Tensor, sigmas: Tensor, callback=None, step_cb=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
        denoised = model(x, sigmas[i])
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        d = (x - denoised) / _f(sigmas[i])
        x = x + d * _f(sigmas[i + 1] - sigmas[i])
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
def sample_heun(model, x: Tensor, sigmas: Tensor, callback=None, step_cb=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
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
def sample_dpm_2(model, x: Tensor, sigmas: Tensor, callback=None, step_cb=None, **_):
    sigmas = sigmas.to("cpu", torch.float32)
    for i in range(len(sigmas) - 1):
        if step_cb is not None:
            step_cb(i)
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
                      strength: Optional[float] = None, config: SchedulerConfig = SchedulerConfig()):
        if self.eps_unet is None:
            raise ValueError("Epsilon unet needs to be set before timesteps")
        s = self.schedule
        self.unets = [KDiffusionUNetWrapper(e, s) for e in self.eps_unets]
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
        if self.eta is not None:
            kwargs["eta"] = self.eta
        kwargs["noise_sampler"] = lambda _, __: batched_randn(latents.shape, self.generators, self.device, self.dtype)
        model = self.unet
        if k_wrap is not None or k_model is not None:
            u_off = self.start_offset / len(self.sigmas)
            state = {"i": 0, "i_max": len(sigmas) - 1}

            def tracked(x, sigma):
                u = u_off + (1 - u_off) * state["i"] / state["i_max"]
                u = max(min(u, 0.999), 0)
                if k_model is not None:
                    return k_model(x, sigma, u)
                return k_wrap(self.unet(x, sigma), u)

            model = tracked
            kwargs["step_cb"] = lambda i: state.__setitem__("i", i)
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


SAMPLERS.update({
    "dpm_2_a": (sample_dpm_2_ancestral, {}),
    "lms": (sample_lms, {}),
})


# ------------------------------------------------------------------------------
# diffusers-scheduler loop (reference common_scheduler.py:179-314: DDIM, PLMS = PNDM(skip_prk_steps=True)).
# The step arithmetic is diffusers' [3P, ~=0.16.0, not vendored]; restated from the published algorithms with the
# SD1.x scheduler_config.json values (scaled_linear betas, steps_offset=1, set_alpha_to_one=False, no clipping).
# Coefficients are host fp32/64 scalars applied to the device latents (no per-step sync).
# ------------------------------------------------------------------------------
class DiffusersLikeScheduler:
    init_noise_sigma = 1.0

    def __init__(self, kind: str, num_train_timesteps: int = 1000, steps_offset: int = 1):
        if kind not in ("ddim", "plms"):
            raise NotImplementedError(kind)
        self.kind, self.T, self.steps_offset = kind, num_train_timesteps, steps_offset
        self.alphas_cumprod = DiscreteSchedule(num_train_timesteps).alphas_cumprod.double()
        self.final_alpha_cumprod = self.alphas_cumprod[0]  # set_alpha_to_one = False

    def set_timesteps(self, n: int):
        import numpy as np
        self.n = n
        ratio = self.T // n
        base = (np.arange(0, n) * ratio).round().astype(np.int64) + self.steps_offset
        if self.kind == "ddim":
            self.timesteps = base[::-1].copy()
        else:  # PNDM with skip_prk_steps: the second timestep is visited twice
            self.timesteps = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.ets, self.counter, self.cur_sample = [], 0, None

    def _ac(self, t: int) -> float:
        return float(self.alphas_cumprod[t]) if t >= 0 else float(self.final_alpha_cumprod)

    def step(self, eps: Tensor, t: int, sample: Tensor) -> Tensor:
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
        self.eps_unet, self.unet = None, None
        self.start_offset = 0

    def set_eps_unet(self, eps_unet):
        self.eps_unet = eps_unet

    def set_timesteps(self, num_inference_steps: int, start_offset=None, strength=None, config=None):
        if self.eps_unet is None:
            raise ValueError("Epsilon unet needs to be set before timesteps")
        self.sched.set_timesteps(num_inference_steps)
        if strength is not None:
            init = min(int(num_inference_steps * strength), num_inference_steps)
            self.start_offset = max(num_inference_steps - init, 0)
        else:
            self.start_offset = start_offset or 0
        self.unet = type("EvalCounter", (), {"evals": 0})()

    def prepare_initial_latents(self, latents: Tensor) -> Tensor:
        return latents * self.sched.init_noise_sigma

    def add_noise(self, latents: Tensor, noise: Tensor) -> Tensor:
        a = float(self.sched.alphas_cumprod[int(self.sched.timesteps[self.start_offset])])
        return a ** 0.5 * latents + (1 - a) ** 0.5 * noise

    def add_noise_at(self, latents: Tensor, noise: Tensor, t: int) -> Tensor:
        a = float(self.sched.alphas_cumprod[t])
        return a ** 0.5 * latents + (1 - a) ** 0.5 * noise

    def loop(self, latents: Tensor, callback=None, d_wrap=None) -> Tensor:
        """d_wrap(xt, t, u) -> xt post-processes each new sample (reference Mode.wrap_d_unet)."""
        x = latents
        ts = self.sched.timesteps[self.start_offset:]
        u_off = self.start_offset / max(len(self.sched.timesteps), 1)
        for i, t in enumerate(ts):
            eps = self.eps_unet(x, int(t))
            self.unet.evals += 1
            x = self.sched.step(eps, int(t), x)
            if d_wrap is not None:
                u = u_off + (1 - u_off) * i / max(len(ts), 1)
                x = d_wrap(x, int(t), max(min(u, 0.999), 0))
            if callback is not None:
                callback({"x": x, "i": i, "t": int(t)})
        return x


DIFFUSERS_SAMPLERS = ("ddim", "plms")


def make_scheduler(sampler, generators, device, dtype=torch.float32):
    """sampler name -> loop driver, the split the reference makes in build_sampler_set (samplers.py:110-130)."""
    if isinstance(sampler, str) and sampler in DIFFUSERS_SAMPLERS:
        return DiffusersScheduler(sampler, generators, device, dtype)
    return KDiffusionScheduler(sampler, generators, device, dtype)
