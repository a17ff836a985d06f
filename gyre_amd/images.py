"""Post-decode image helpers (host/device PyTorch): the outmask composite of the reference's pipeline tail.

reference: gyre/pipeline/unified_pipeline.py:2493-2510 (composite), gyre/images.py:667-672 + :49-82 (8-bit round
trip), gyre/match_histograms.py:12-36,84-92 (per-channel CDF matching, statistics taken over the whole batch - this
step couples the images of a batch in the reference, too).
"""
from __future__ import annotations

import numpy as np
import torch

Tensor = torch.Tensor


def _lut_from_counts(src_counts: np.ndarray, tmpl_counts: np.ndarray) -> np.ndarray:
    """256-entry map value -> matched value for one channel (unsigned-integer branch of _match_cumulative_cdf)."""
    tmpl_values = np.nonzero(tmpl_counts)[0]
    tmpl_nz = tmpl_counts[tmpl_values]
    src_q = np.cumsum(src_counts) / src_counts.sum()
    tmpl_q = np.cumsum(tmpl_nz) / tmpl_counts.sum()
    return np.interp(src_q, tmpl_q, tmpl_values)


def match_histograms_u8(image: Tensor, reference: Tensor) -> Tensor:
    """image, reference: uint8 [B,H,W,C] on any device -> uint8 matched image (float results truncated, as numpy's
    assignment into the uint8 output array does).  Only the 256-bin counts and the LUT cross the PCIe bus."""
    if image.ndim != reference.ndim:
        raise ValueError("Image and reference must have the same number of channels.")
    if image.shape[-1] != reference.shape[-1]:
        raise ValueError("Number of channels in the input image and reference image must match!")
    out = torch.empty_like(image)
    for c in range(image.shape[-1]):
        src = image[..., c].reshape(-1).to(torch.int64)
        sc = torch.bincount(src, minlength=256).cpu().numpy()
        tc = torch.bincount(reference[..., c].reshape(-1).to(torch.int64), minlength=256).cpu().numpy()
        lut = torch.from_numpy(_lut_from_counts(sc, tc).astype(np.uint8)).to(image.device)
        out[..., c] = lut[src].reshape(image.shape[:-1])
    return out


def match_histograms(image: Tensor, reference: Tensor) -> Tensor:
    """float BCHW in [0,1] -> float BCHW, through the same 8-bit quantisation the reference applies (toCV / fromCV;
    the channel reordering there is its own inverse and the matching is per channel, so it is skipped)."""
    to_u8 = lambda t: (t.to(torch.float32) * 255).round().to(torch.uint8).permute(0, 2, 3, 1)
    m = match_histograms_u8(to_u8(image), to_u8(reference))
    return (m.permute(0, 3, 1, 2).to(torch.float32) / 255.0).to(image)


def outmask_composite(result: Tensor, source: Tensor, outmask: Tensor) -> Tensor:
    """result [B,3,H,W] in [0,1]; source / outmask [1 or B, >=3, H, W].  The generated pixels are tone-matched to the
    source-with-result composite, then pasted over the source where outmask is 1."""
    B = result.shape[0]
    rep = lambda t: (torch.cat([t] * B) if t.shape[0] == 1 else t)[:, [0, 1, 2]].to(result)
    outmask, source = rep(outmask), rep(source)
    reference = source * (1 - outmask) + result * outmask
    matched = match_histograms(result, reference)
    return source * (1 - outmask) + matched * outmask
