"""Hires fix and UNet grafting: blends of two denoisers during the sampler loop (host PyTorch).

Restates the reference's
  Easing            gyre/pipeline/easing.py:21-46  (curves from the absent ``easing_functions`` package: standard
                                                    Penner ease-in-out formulas, restated - parity unpinned)
  GraftUnets        gyre/pipeline/unet/graft.py:16-56
  HiresUnetWrapper  gyre/pipeline/unet/hires_fix.py:21-235
for the k-diffusion denoiser signature ``unet(x, sigma, u) -> x0_hat`` (u = progress in [0, 0.999]).

Hires fix: the sampler state is a batch of 2B latents - the natural-size (UNet training size, 64x64) image embedded
in the centre of a full-size canvas, and the full-size image.  Both are denoised; early in the run (p = cubic ease of
u over [0, 0.667]) the two predictions are exchanged through lanczos2 resampling and a per-element random mask drawn
from the per-image generators, so that the composition of the full-size image is laid out at the resolution the
UNet was trained on.  From p >= 0.999 on only the full-size half is evaluated.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence

import torch

from .resize import resize_lanczos2, resize_nearest
from .schedulers import batched_rand

Tensor = torch.Tensor


# -- easing ---------------------------------------------------------------------------------------------------
def _cubic(t): return 4 * t * t * t if t < 0.5 else 0.5 * (2 * t - 2) ** 3 + 1
def _quad(t): return 2 * t * t if t < 0.5 else -2 * t * t + 4 * t - 1
def _quartic(t): return 8 * t ** 4 if t < 0.5 else -8 * (t - 1) ** 4 + 1
def _quintic(t): return 16 * t ** 5 if t < 0.5 else 0.5 * (2 * t - 2) ** 5 + 1
def _sine(t): return 0.5 * (1 - math.cos(t * math.pi))
def _circular(t): return 0.5 * (1 - math.sqrt(1 - 4 * t * t)) if t < 0.5 else 0.5 * (math.sqrt(-(2 * t - 3) * (2 * t - 1)) + 1)


def _expo(t):
    if t == 0 or t == 1:
        return t
    return 0.5 * 2 ** (20 * t - 10) if t < 0.5 else -0.5 * 2 ** (-20 * t + 10) + 1


EASINGS = {"linear": lambda t: t, "quad": _quad, "cubic": _cubic, "quartic": _quartic, "quintic": _quintic,
           "sine": _sine, "circular": _circular, "expo": _expo}


class Easing:
    """interp(u): floor below start, 1 above end, floor + (1 - floor) * ease((u - start) / (end - start)) between."""

    def __init__(self, floor: float, start: float, end: float, easing: str):
        if easing not in EASINGS:
            raise ValueError(f"unknown easing {easing!r} (have {sorted(EASINGS)})")
        self.floor, self.start, self.end = floor, start, end
        self.fn = EASINGS[easing]

    def interp(self, u: float) -> float:
        if u < self.start:
            return self.floor
        if u > self.end:
            return 1
        a = self.fn((u - self.start) / (self.end - self.start))
        return self.floor + (1 - self.floor) * a


# -- graft ----------------------------------------------------------------------------------------------------
class GraftUnets:
    """Per-element stochastic blend from unet_root to unet_top while p(u) goes 0 -> 1 (default sine over [0.1, 0.3])."""

    def __init__(self, unet_root: Callable, unet_top: Callable, generators: List[torch.Generator], blend: Optional[dict] = None):
        self.unet_root, self.unet_top, self.generators = unet_root, unet_top, generators
        self.easing = Easing(**{"floor": 0, "start": 0.1, "end": 0.3, "easing": "sine", **(blend or {})})

    def __call__(self, latents: Tensor, step, u: float) -> Tensor:
        p = self.easing.interp(u)
        if p <= 0:
            return self.unet_root(latents, step, u)
        if p >= 1:
            return self.unet_top(latents, step, u)
        root = self.unet_root(latents, step, u)
        top = self.unet_top(latents, step, u)
        randmap = batched_rand(top.shape, self.generators, top.device, top.dtype)
        return torch.where(randmap >= p, root, top)

    @staticmethod
    def merge_initial_latents(left: Tensor, right: Tensor) -> Tensor:
        return left

    @staticmethod
    def split_result(left: Tensor, right: Tensor) -> Tensor:
        return right


# -- hires fix ------------------------------------------------------------------------------------------------
def scale_into(latents: Tensor, scale: float, target: Optional[Tensor] = None,
               target_shape: Optional[Sequence[int]] = None, mode: str = "lanczos") -> Tensor:
    """Resample by ``scale`` then centre-crop to / centre-place inside the target (replicate-padding the border when
    only a shape is given).  hires_fix.py:43-89."""
    latents = resize_nearest(latents, scale) if mode == "nearest" else resize_lanczos2(latents, scale)
    if (target is None) == (target_shape is None):
        raise ValueError("exactly one of target or target_shape is required")
    if target_shape is None:
        target_shape = target.shape
    th, tw = target_shape[-2], target_shape[-1]
    offh, offw = (th - latents.shape[-2]) // 2, (tw - latents.shape[-1]) // 2
    if offh < 0:
        latents = latents[:, :, -offh:-offh + th, :]
        offh = 0
    if offw < 0:
        latents = latents[:, :, :, -offw:-offw + tw]
        offw = 0
    if target is not None:
        target[:, :, offh:offh + latents.shape[-2], offw:offw + latents.shape[-1]] = latents
        return target
    pad = (offw, tw - latents.shape[-1] - offw, offh, th - latents.shape[-2] - offh)
    return torch.nn.functional.pad(latents, pad, mode="replicate")


def down_scale_factor(src_shape, target_shape, oos_fraction: float) -> float:
    """oos ("out of square") fraction 1 fits the whole source inside the target, 0 fills the target (crop)."""
    sh, sw = target_shape[-2] / src_shape[-2], target_shape[-1] / src_shape[-1]
    return min(sh, sw) * oos_fraction + max(sh, sw) * (1 - oos_fraction)


def up_scale_factor(src_shape, target_shape, oos_fraction: float) -> float:
    return 1 / down_scale_factor(target_shape, src_shape, oos_fraction)


def image_to_natural(natural_size: int, image: Tensor, oos_fraction: float) -> Tensor:
    """Downscale a full-size conditioning image / mask to the UNet's natural pixel size (hires_fix.py:205-215)."""
    shape = [natural_size, natural_size]
    return scale_into(image, down_scale_factor(image.shape, shape, oos_fraction), target_shape=shape)


class HiresUnetWrapper:
    def __init__(self, unet_natural: Callable, unet_hires: Callable, generators: List[torch.Generator],
                 natural_size: Sequence[int], oos_fraction: float):
        self.unet_natural, self.unet_hires, self.generators = unet_natural, unet_hires, generators
        self.natural_size, self.oos_fraction = tuple(natural_size), oos_fraction
        self.easing = Easing(floor=0, start=0, end=0.667, easing="cubic")

    def __call__(self, latents: Tensor, step, u: float) -> Tensor:
        p = self.easing.interp(u)
        lo_in, hi_in = latents.chunk(2)
        if isinstance(step, torch.Tensor) and step.ndim > 0 and step.shape[0] > 1:
            lo_t, hi_t = step.chunk(2)
        else:
            lo_t = hi_t = step
        hi = self.unet_hires(hi_in, hi_t, u)
        if p >= 0.999:                                   # past the exchange stage: only the full-size half matters
            return torch.cat([lo_in, hi])
        h, w = latents.shape[-2], latents.shape[-1]
        th, tw = self.natural_size
        oh, ow = (h - th) // 2, (w - tw) // 2
        lo = self.unet_natural(lo_in[:, :, oh:oh + th, ow:ow + tw], lo_t, u)
        # full-size prediction, downscaled, replaces a fraction p of the natural-size prediction ...
        hi_down = scale_into(hi, down_scale_factor(hi.shape, lo.shape, self.oos_fraction), target_shape=lo.shape)
        randmap = batched_rand(lo.shape, self.generators, lo.device, lo.dtype)
        lo_merged = torch.where(randmap >= p, lo, hi_down)
        # ... and the natural-size prediction, upscaled into a copy of the full-size one, replaces 1 - p of it
        lo_up = scale_into(lo, up_scale_factor(lo.shape, hi.shape, self.oos_fraction), target=hi.clone())
        randmap = batched_rand(hi.shape, self.generators, hi.device, hi.dtype)
        hi_merged = torch.where(randmap >= p, lo_up, hi)
        lo_full = torch.zeros_like(hi_merged)
        lo_full[:, :, oh:oh + th, ow:ow + tw] = lo_merged
        return torch.cat([lo_full, hi_merged])

    @staticmethod
    def merge_initial_latents(left: Tensor, right: Tensor) -> Tensor:
        canvas = torch.zeros_like(right)
        th, tw = left.shape[-2], left.shape[-1]
        oh, ow = (right.shape[-2] - th) // 2, (right.shape[-1] - tw) // 2
        canvas[:, :, oh:oh + th, ow:ow + tw] = left
        return torch.cat([canvas, right])

    @staticmethod
    def split_result(left: Tensor, right: Tensor) -> Tensor:
        return right.chunk(2)[1]
