"""Token merging (ToMe) for self-attention keys / values - CPU restatement.  TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED: the reference only CALLS the algorithm (nonfree/tome_unet.py:138-182: ``merge, _ =
bipartite_soft_matching(key, r, class_token, distill_token)``, ``key = merge_wavg(merge, key)``, ``value =
merge_wavg(merge, value)``); it lives in the facebookresearch/ToMe git submodule (.gitmodules:4-6), which is absent from the
reference tree, and the reference disables the feature ("isn't finished", unified_pipeline.py:1580-1588).  Restated from
the published method (Bolya et al. 2023, tome/merge.py) with class / distill tokens off, as the reference passes them:

    metric = key / key.norm(dim=-1, keepdim=True)
    a, b = metric[..., ::2, :], metric[..., 1::2, :]
    scores = a @ b.transpose(-1, -2)
    node_max, node_idx = scores.max(dim=-1)
    edge_idx = node_max.argsort(dim=-1, descending=True)
    unm_idx, src_idx = edge_idx[..., r:], edge_idx[..., :r] ;  dst_idx = node_idx.gather(src_idx)
    merge(x): cat([x_a[unm_idx], x_b.scatter_reduce(dst_idx, x_a[src_idx], "sum")])      merge_wavg: ... / token counts

Tie rules the published code leaves to torch.max / torch.argsort are fixed as the native kernels fix them: first maximal
index; equal scores rank by ascending token index.  ``emulate_bf16`` applies the two roundings of the native path
(normalised keys and similarity scores are stored as bf16) so that the SELECTIONS can be compared, not only the effect."""
from __future__ import annotations

import torch

Tensor = torch.Tensor


def _bf16(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def bipartite_soft_matching(key: Tensor, r: int, emulate_bf16: bool = True):
    """key [B, N, C] -> (order [B, N//2] a-token indices by descending best-match score, node_idx [B, N//2], r_eff)."""
    B, N, C = key.shape
    half = N // 2
    r = max(0, min(int(r), half))
    metric = key.float() / key.float().norm(dim=-1, keepdim=True).clamp_min(1e-15)
    if emulate_bf16:
        metric = _bf16(metric)
    a, b = metric[:, 0:2 * half:2], metric[:, 1:2 * half:2]
    scores = a @ b.transpose(-1, -2)
    if emulate_bf16:
        scores = _bf16(scores)
    node_max, node_idx = scores.max(dim=-1)                                  # first maximal index (CPU semantics)
    order = torch.sort(-node_max, dim=-1, stable=True).indices                # descending score, ascending index
    return order, node_idx, r


def merge_wavg(x: Tensor, order: Tensor, node_idx: Tensor, r: int) -> Tensor:
    """x [B, N, C] -> [B, N - r, C]: unmerged a tokens (rank order), then the b tokens averaged with what was merged into
    them, then a trailing unpaired token when N is odd."""
    B, N, C = x.shape
    half = N // 2
    xa, xb = x[:, 0:2 * half:2].float(), x[:, 1:2 * half:2].float()
    out = []
    for bi in range(B):
        src = order[bi, :r]
        unm = order[bi, r:]
        dst = xb[bi].clone()
        cnt = torch.ones(half, 1)
        dst.index_add_(0, node_idx[bi, src], xa[bi, src])
        cnt.index_add_(0, node_idx[bi, src], torch.ones(len(src), 1))
        out.append(torch.cat([xa[bi, unm], dst / cnt, x[bi, 2 * half:].float()], dim=0))
    return torch.stack(out)


def tome_merge_kv(key: Tensor, value: Tensor, r: int, emulate_bf16: bool = True):
    order, node_idx, r = bipartite_soft_matching(key, r, emulate_bf16)
    if r == 0:
        return key.float(), value.float()
    return merge_wavg(key, order, node_idx, r), merge_wavg(value, order, node_idx, r)
