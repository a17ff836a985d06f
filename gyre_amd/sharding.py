"""Data-parallel sharding of one request batch across the GPUs of a node.

The reference only parallelises at request granularity (a queue of device slots,
gyre/manager.py:648-651,2106-2141; sub-batches of one request run sequentially,
gyre/services/generate.py:1049-1091).  Here one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI) takes a contiguous slice of the images - same split rule as the
reference's batched_seeds (generate.py:977-990: even split, remainder spread over the first
ranks) - runs the whole denoising loop on its slice with its own per-image generators, and
the only collective is one all_gather of the finished latents (32 KB / image at 512^2).
Weights are replicated.  Because every random draw is per image (randtools.py:39-64) and no
kernel's arithmetic depends on batch position, an image's result does not depend on which rank or
batch slot it lands in (bit-identical for equal shard sizes; across different shard sizes the GEMM
planner may pick another split-K factor, which changes results only at bf16-rounding level).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, end) per rank; sizes differ by at most one, larger shards first."""
    if total < 0 or world < 1:
        raise ValueError("total >= 0 and world >= 1 required")
    base, rem = divmod(total, world)
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((s, s + n))
        s += n
    return out


def gather_batches(local: torch.Tensor, sizes: Sequence[int], group=None) -> torch.Tensor:
    """all_gather of per-rank [b_r, ...] tensors with ragged b_r (padded to the largest shard)."""
    world = dist.get_world_size(group)
    mx = max(sizes)
    # gloo gathers host tensors only: stage through the CPU there (tests / single-GPU development; RCCL takes the device path)
    via_host = local.is_cuda and dist.get_backend(group) == "gloo"
    src = local.cpu() if via_host else local
    pad = torch.zeros((mx, *src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[: src.shape[0]] = src
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    out = torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
    return out.to(local.device) if via_host else out


def generate_sharded(pipe, *, seeds: Sequence[int], text_embeddings: torch.Tensor,
                     uncond_embeddings: Optional[torch.Tensor] = None, group=None, gather: bool = True,
                     bit_exact: bool = False, **kw):
    """Run pipe(...) on this rank's slice of the batch and gather the finished latents.

    bit_exact=True switches the native library to batch-invariant planning (modules.set_batch_invariant) for the
    call, so the gathered result is bit-identical to a single-GPU run of the whole batch for ANY world size, also
    when the shards differ in size; off, that only holds between equal-sized shards.

    Returns (latents_full [B,4,h,w] on every rank, (start, end) of the local slice).
    Ranks with an empty slice (more GPUs than images) only take part in the collective."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = shard_bounds(len(seeds), world)
    s, e = bounds[rank]
    kw = dict(kw)
    kw["output_type"] = "latent"
    if e > s:
        te = text_embeddings if text_embeddings.shape[0] == 1 else text_embeddings[s:e]
        ue = uncond_embeddings
        if ue is not None and ue.shape[0] != 1:
            ue = ue[s:e]
        for key in ("added_cond", "uncond_added_cond"):     # per-image SDXL conditioning follows the slice
            if kw.get(key) is not None:
                kw[key] = {k: (v if v.shape[0] == 1 else v[s:e]) for k, v in kw[key].items()}
        prev = None
        if bit_exact:
            from .modules import set_batch_invariant
            prev = set_batch_invariant(16)
        try:
            local = pipe(seeds=list(seeds[s:e]), text_embeddings=te, uncond_embeddings=ue, **kw)
        finally:
            if prev is not None:
                set_batch_invariant(prev)
    else:
        h, w = kw.get("height", 512) // 8, kw.get("width", 512) // 8
        local = torch.zeros((0, 4, h, w), dtype=torch.float32, device=pipe.device)
    if not gather:
        return local, (s, e)
    return gather_batches(local, [b - a for a, b in bounds], group), (s, e)
