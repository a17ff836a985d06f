"""Lanczos-2 resampling of latents / images by an arbitrary scale factor (host PyTorch).

The reference resamples between the natural-size and the full-size latents of its hires fix with
``resize_right.resize(x, scale_factors=s, interp_method=lanczos2, pad_mode="replicate", antialiasing=False)``
(gyre/pipeline/unet/hires_fix.py:46-52).  ResizeRight is an un-vendored git submodule
(.gitmodules:10-12, ``gyre/src/ResizeRight`` is empty in the reference tree), so this file restates the published
algorithm (Shocher, "ResizeRight", 2021) - *parity unpinned*:

  out_size            = ceil(in_size * scale)
  projected grid      p_i = i / scale + (in_size - 1) / 2 - (out_size - 1) / (2 * scale)     (centres aligned)
  field of view       left_i = ceil(p_i - support / 2 - eps), taps left_i .. left_i + support - 1   (support = 4)
  weights             w_ij = lanczos2(p_i - j), normalised to sum 1 per output sample
  boundary            taps outside the input are clamped to the edge sample (replicate padding)

Without antialiasing the kernel is not stretched when downscaling (that is what the reference asks for).
The separable resize is two small dense matrix products PER SAMPLE: a batched matmul may pick its kernel (and with it
the fp32 summation order) by batch count, which would break the batch-independence property (reference
tests/batch_independance.py:15-27) under the hires fix; one call per image has the same shape whatever the batch.
"""
from __future__ import annotations

import math
from functools import lru_cache

import torch

_EPS = float(torch.finfo(torch.float32).eps)


def lanczos2(x: torch.Tensor) -> torch.Tensor:
    """sinc(x) sinc(x/2) on |x| < 2, written with the same eps guard as the published code so x = 0 gives 1."""
    pix = math.pi * x
    return ((torch.sin(pix) * torch.sin(pix / 2) + _EPS) / ((pix * pix / 2) + _EPS)) * (x.abs() < 2)


@lru_cache(maxsize=64)
def _weight_matrix(in_size: int, scale: float) -> torch.Tensor:
    """[out_size, in_size] float64 resampling matrix (rows sum to 1)."""
    out_size = int(math.ceil(in_size * scale))
    support = 4
    i = torch.arange(out_size, dtype=torch.float64)
    proj = i / scale + (in_size - 1) / 2 - (out_size - 1) / (2 * scale)
    left = torch.ceil(proj - support / 2 - _EPS).to(torch.int64)
    taps = left[:, None] + torch.arange(support)[None, :]                      # [out, 4] input coordinates
    w = lanczos2(proj[:, None] - taps.to(torch.float64))
    s = w.sum(dim=1, keepdim=True)
    s[s == 0] = 1
    w = w / s
    m = torch.zeros(out_size, in_size, dtype=torch.float64)
    m.scatter_add_(1, taps.clamp(0, in_size - 1), w)                           # replicate padding = clamp
    return m


_DEVICE_CACHE: dict = {}


def _device_matrices(in_size: int, scale: float, device):
    """(matrix, its transpose) as fp32 ON `device`, cached: the hires fix resamples the same two sizes on every step of a request, and
    a host-built matrix costs a pageable host-to-device copy - a stream sync that drains the launch queue - per call (measured:
    1.5 s of a 2.5 s inpaint-768 step were spent blocked in those copies)."""
    key = (in_size, scale, str(device))
    hit = _DEVICE_CACHE.get(key)
    if hit is None:
        if len(_DEVICE_CACHE) > 256:
            _DEVICE_CACHE.clear()
        m = _weight_matrix(in_size, scale).to(device, torch.float32)
        hit = _DEVICE_CACHE[key] = (m, m.t().contiguous())
    return hit


def resize_lanczos2(x: torch.Tensor, scale: float) -> torch.Tensor:
    """Resize the last two dims of x by ``scale`` (same factor both ways), lanczos2, replicate pad, no antialias."""
    if scale <= 0:
        raise ValueError("scale must be positive")
    if scale == 1:
        return x.clone()
    h, w = x.shape[-2], x.shape[-1]
    mh, _ = _device_matrices(h, float(scale), x.device)
    _, mwt = _device_matrices(w, float(scale), x.device)
    xf = x.to(torch.float32)
    if xf.ndim < 4:
        return torch.matmul(torch.matmul(mh, xf), mwt).to(x.dtype)
    lead = xf.shape[:-3]
    xs = xf.reshape((-1,) + tuple(xf.shape[-3:]))
    ys = [torch.matmul(torch.matmul(mh, s_), mwt) for s_ in xs]               # rows, then columns, one sample at a time
    y = torch.stack(ys).reshape(lead + ys[0].shape)
    return y.to(x.dtype)


def cubic(x: torch.Tensor) -> torch.Tensor:
    """Keys cubic convolution kernel (a = -0.5), support 4 - ResizeRight's default interpolation method."""
    ax = x.abs()
    ax2, ax3 = ax * ax, ax * ax * ax
    return (1.5 * ax3 - 2.5 * ax2 + 1) * (ax <= 1) + (-0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2) * ((1 < ax) & (ax <= 2))


_GENERAL_CACHE: dict = {}


def _weight_matrix_general(in_size: int, out_size: int, scale: float, method: str, antialiasing: bool,
                           pad_mode: str, device="cpu") -> torch.Tensor:
    """[out_size, in_size] float32 matrix of ResizeRight's 1-D resampling for one axis (same published algorithm as
    _weight_matrix, with the pieces the CLIP cut-outs use: an explicit output size, the antialiasing stretch of the kernel
    when downscaling - kernel(x) -> scale * kernel(scale * x), support / scale - and reflect / constant / replicate
    boundary handling folded into the matrix).  Built in float64 ON the target device: the CLIP cut-outs ask for a new
    (random) size every time, and a host-built matrix would cost a pageable host-to-device copy - a stream sync - per cut-out."""
    device = torch.device(device)
    key = (in_size, out_size, float(scale), method, bool(antialiasing), pad_mode, str(device))
    hit = _GENERAL_CACHE.get(key)
    if hit is not None:
        return hit
    kernel, support = {"cubic": (cubic, 4.0), "lanczos2": (lanczos2, 4.0)}[method]
    if scale < 1 and antialiasing:
        base, k_support = kernel, support / scale
        kern = lambda a: scale * base(scale * a)
    else:
        kern, k_support = kernel, support
    i = torch.arange(out_size, dtype=torch.float64, device=device)
    proj = i / scale + (in_size - 1) / 2 - (out_size - 1) / (2 * scale)
    left = torch.ceil(proj - k_support / 2 - _EPS).to(torch.int64)
    ntaps = int(math.ceil(k_support - _EPS))
    taps = left[:, None] + torch.arange(ntaps, device=device)[None, :]
    w = kern(proj[:, None] - taps.to(torch.float64))
    s_ = w.sum(dim=1, keepdim=True)
    s_ = torch.where(s_ == 0, torch.ones_like(s_), s_)
    w = w / s_
    m = torch.zeros(out_size, in_size, dtype=torch.float64, device=device)
    # extreme taps from the closed form (no device read-back): first / last row reach furthest
    p_first = (in_size - 1) / 2 - (out_size - 1) / (2 * scale)
    p_last = (out_size - 1) / scale + p_first
    lo = math.ceil(p_first - k_support / 2 - _EPS)
    hi = math.ceil(p_last - k_support / 2 - _EPS) + ntaps - 1
    if pad_mode == "replicate":
        m.scatter_add_(1, taps.clamp(0, in_size - 1), w)
    elif pad_mode == "reflect":                       # torch 'reflect': mirror without repeating the edge sample
        if lo < -(in_size - 1) or hi > 2 * (in_size - 1):
            raise ValueError("reflect padding needs the padding to be smaller than the input")
        idx = taps.abs()
        idx = torch.where(idx > in_size - 1, 2 * (in_size - 1) - idx, idx)
        m.scatter_add_(1, idx, w)
    elif pad_mode == "constant":                      # zeros outside
        inside = (taps >= 0) & (taps < in_size)
        m.scatter_add_(1, taps.clamp(0, in_size - 1), w * inside)
    else:
        raise ValueError(f"pad_mode {pad_mode!r}")
    m = m.to(torch.float32)
    if len(_GENERAL_CACHE) > 512:
        _GENERAL_CACHE.clear()
    _GENERAL_CACHE[key] = m
    return m


def resize_right(x: torch.Tensor, out_shape=None, scale_factors=None, interp_method: str = "cubic",
                 antialiasing: bool = True, pad_mode: str = "constant") -> torch.Tensor:
    """``resize_right.resize(x, scale_factors=None, out_shape=(h, w), pad_mode=...)`` on the last two axes - the call
    the reference makes for its CLIP cut-outs (gyre/pipeline/unet/clipguided.py:81-83, 362-364).  Differentiable (two
    matrix products per sample); un-vendored upstream, *parity unpinned* like resize_lanczos2."""
    h, w = x.shape[-2], x.shape[-1]
    if out_shape is not None:
        oh, ow = int(out_shape[-2]), int(out_shape[-1])
        sh, sw = oh / h, ow / w
    elif scale_factors is not None:
        sh = sw = float(scale_factors)
        oh, ow = int(math.ceil(h * sh)), int(math.ceil(w * sw))
    else:
        raise ValueError("pass out_shape or scale_factors")
    mh = _weight_matrix_general(h, oh, float(sh), interp_method, bool(antialiasing), pad_mode, x.device)
    mwt = _weight_matrix_general(w, ow, float(sw), interp_method, bool(antialiasing), pad_mode, x.device).t().contiguous()
    xf = x.to(torch.float32)
    if xf.ndim < 4:
        return torch.matmul(torch.matmul(mh, xf), mwt).to(x.dtype)
    lead = xf.shape[:-3]
    xs = xf.reshape((-1,) + tuple(xf.shape[-3:]))
    ys = [torch.matmul(torch.matmul(mh, s_), mwt) for s_ in xs]
    return torch.stack(ys).reshape(lead + ys[0].shape).to(x.dtype)


def resize_nearest(x: torch.Tensor, scale: float) -> torch.Tensor:
    hs, ws = int(x.shape[-2] * scale), int(x.shape[-1] * scale)
    return torch.nn.functional.interpolate(x, size=(hs, ws), mode="nearest")
