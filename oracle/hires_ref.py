"""ORACLE (test infrastructure only - never imported by the product): independent CPU restatement of the reference's
hires-fix / graft arithmetic, written as explicit loops so that it shares no code with gyre_amd/resize.py + hires.py.

Follows
  gyre/pipeline/unet/hires_fix.py:43-89 (scale_into), :92-100 (scale factors), :123-203 (HiresUnetWrapper.__call__),
  :217-235 (merge / split), gyre/pipeline/unet/graft.py:32-48, gyre/pipeline/easing.py:21-46.
PARITY UNPINNED for the two third-party pieces the reference imports but does not vendor: ResizeRight
(.gitmodules:10-12, empty directory) and easing_functions (pyproject dependency, absent); both are restated from their
published definitions (lanczos2 = sinc(x) sinc(x/2) on |x|<2, centre-aligned grid, 4 taps, replicate border, weights
normalised; Penner ease-in-out curves).
"""
import math

import numpy as np
import torch

EPS = float(np.finfo(np.float32).eps)


def _lanczos2(x: float) -> float:
    if abs(x) >= 2:
        return 0.0
    px = math.pi * x
    return (math.sin(px) * math.sin(px / 2) + EPS) / (px * px / 2 + EPS)


def resize_1d_ref(v: np.ndarray, scale: float) -> np.ndarray:
    """v: [n] -> [ceil(n*scale)]"""
    n = v.shape[0]
    m = int(math.ceil(n * scale))
    out = np.zeros(m, dtype=np.float64)
    for i in range(m):
        p = i / scale + (n - 1) / 2 - (m - 1) / (2 * scale)
        left = int(math.ceil(p - 2 - EPS))
        ws = [_lanczos2(p - (left + k)) for k in range(4)]
        tot = sum(ws) or 1.0
        acc = 0.0
        for k in range(4):
            j = min(max(left + k, 0), n - 1)
            acc += ws[k] / tot * float(v[j])
        out[i] = acc
    return out


def resize_lanczos2_ref(x: torch.Tensor, scale: float) -> torch.Tensor:
    a = x.detach().cpu().double().numpy()
    lead = a.shape[:-2]
    a2 = a.reshape(-1, a.shape[-2], a.shape[-1])
    rows = np.stack([np.stack([resize_1d_ref(img[:, c], scale) for c in range(img.shape[1])], axis=1) for img in a2])
    out = np.stack([np.stack([resize_1d_ref(img[r, :], scale) for r in range(img.shape[0])], axis=0) for img in rows])
    return torch.from_numpy(out.reshape(*lead, out.shape[-2], out.shape[-1])).to(x.dtype)


def ease_ref(kind: str, floor: float, start: float, end: float, u: float) -> float:
    if u < start:
        return floor
    if u > end:
        return 1
    t = (u - start) / (end - start)
    if kind == "cubic":
        a = 4 * t ** 3 if t < 0.5 else 1 - ((-2 * t + 2) ** 3) / 2
    elif kind == "sine":
        a = -(math.cos(math.pi * t) - 1) / 2
    elif kind == "linear":
        a = t
    else:
        raise NotImplementedError(kind)
    return floor + (1 - floor) * a


def _place(src: torch.Tensor, th: int, tw: int, target=None) -> torch.Tensor:
    """centre crop / centre placement; replicate border when no target canvas is given"""
    offh, offw = (th - src.shape[-2]) // 2, (tw - src.shape[-1]) // 2
    if offh < 0:
        src = src[:, :, -offh:-offh + th]
        offh = 0
    if offw < 0:
        src = src[:, :, :, -offw:-offw + tw]
        offw = 0
    if target is not None:
        target = target.clone()
        target[:, :, offh:offh + src.shape[-2], offw:offw + src.shape[-1]] = src
        return target
    out = torch.zeros(*src.shape[:-2], th, tw, dtype=src.dtype)
    for y in range(th):
        sy = min(max(y - offh, 0), src.shape[-2] - 1)
        for x in range(tw):
            sx = min(max(x - offw, 0), src.shape[-1] - 1)
            out[..., y, x] = src[..., sy, sx]
    return out


def hires_step_ref(unet_natural, unet_hires, latents, step, u, rand_lo, rand_hi, natural_size, oos):
    """One HiresUnetWrapper call; rand_lo / rand_hi are the two uniform maps the reference draws (in that order)."""
    p = ease_ref("cubic", 0, 0, 0.667, u)
    B = latents.shape[0] // 2
    lo_in, hi_in = latents[:B], latents[B:]
    hi = unet_hires(hi_in, step, u)
    if p >= 0.999:
        return torch.cat([lo_in, hi])
    h, w = latents.shape[-2:]
    th, tw = natural_size
    oh, ow = (h - th) // 2, (w - tw) // 2
    lo = unet_natural(lo_in[:, :, oh:oh + th, ow:ow + tw], step, u)
    sdown = min(th / h, tw / w) * oos + max(th / h, tw / w) * (1 - oos)
    hi_down = _place(resize_lanczos2_ref(hi, sdown), th, tw)
    lo_merged = torch.where(rand_lo >= p, lo, hi_down)
    sup = 1 / (min(th / h, tw / w) * oos + max(th / h, tw / w) * (1 - oos))
    lo_up = _place(resize_lanczos2_ref(lo, sup), h, w, target=hi)
    hi_merged = torch.where(rand_hi >= p, lo_up, hi)
    lo_full = torch.zeros_like(hi_merged)
    lo_full[:, :, oh:oh + th, ow:ow + tw] = lo_merged
    return torch.cat([lo_full, hi_merged])
