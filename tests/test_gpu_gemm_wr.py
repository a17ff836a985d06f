"""The W-resident GEMM kernel (tile config 31, kernels_gemm_wr.hip; planner tuning bit 13) against the 8-wave tile kernel on the
square projections it was written for: both sum K in the same order on one accumulator, so every form - plain, residual, row
statistics, folded LayerNorm - has to agree bit for bit with what the planner runs by default.  Replaces the cuBLAS GEMMs behind
torch.nn.Linear in the UNet the reference calls at gyre/pipeline/unet/core.py:274."""
import math

import pytest
import torch

from gpu_util import DEV, randn, repack_bias, repack_linear, st, vp
from gyre_amd import _lib

pytestmark = pytest.mark.gpu


def _both(fn):
    L = _lib.lib()
    arws = torch.empty(640 * 640 * 2, dtype=torch.uint8, device=DEV)
    outs, names = [], []
    for bits in (0, 0x2000):
        L.gyre_debug_set_ar_workspace(vp(arws) if bits else None, arws.numel() if bits else 0)
        L.gyre_debug_gemm_ablation(bits)
        try:
            _lib.prof_enable(None)
            outs.append(fn())
            torch.cuda.synchronize()
            names.append(set(_lib.prof_collect()))
        finally:
            _lib.prof_enable([])
            L.gyre_debug_gemm_ablation(0)
            L.gyre_debug_set_ar_workspace(None, 0)
    assert "k_gemm_wr" in names[1] and "k_gemm_wr" not in names[0], names
    return outs


@pytest.mark.parametrize("M,res", [(4096, False), (8192 + 48, True), (65536, True), (4096 + 16, False)])
def test_linear_and_residual_forms_match_the_tile_kernel(M, res):
    L = _lib.lib()
    K = N = 320
    x = (randn(M, K, seed=1) * 1.3).to(torch.bfloat16).to(DEV)
    w0, b0 = randn(N, K, seed=2) / math.sqrt(K), randn(N, seed=3) * 0.3
    w, b = repack_linear(w0), repack_bias(b0)
    r = randn(M, N, seed=4).to(torch.bfloat16).to(DEV) if res else None

    def run():
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), vp(r), 0, vp(y)))
        return y
    a, c = _both(run)
    assert torch.isfinite(c.float()).all()
    assert torch.equal(a, c)
    ref = x.float() @ w0.to(torch.bfloat16).float().to(DEV).T + b0.to(DEV) + (r.float() if res else 0)
    assert (c.float() - ref).abs().max().item() < 0.06


def test_folded_layernorm_form_matches_the_tile_kernel():
    L = _lib.lib()
    M, K, N = 65536, 320, 320
    x = (randn(M, K, seed=5) * 1.7 + 0.3).to(torch.bfloat16).to(DEV)
    w0, b0 = randn(N, K, seed=6) / math.sqrt(K), randn(N, seed=7) * 0.3
    w, b = repack_linear(w0), repack_bias(b0)
    g, be = (1 + 0.2 * randn(K, seed=8)).to(DEV), (0.1 * randn(K, seed=9)).to(DEV)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(N, K, M), dtype=torch.uint8, device=DEV)

    def run():
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        _lib.check(L.gyre_op_ln_linear(st(), vp(x), M, K, vp(g), vp(be), 1e-5, vp(w), N, vp(b), 0, 0, None, 0, None, 0,
                                       vp(ws), ws.numel(), vp(y)))
        return y
    a, c = _both(run)
    assert torch.equal(a, c)
    xn = torch.nn.functional.layer_norm(x.float(), (K,), g, be, 1e-5)
    ref = xn @ w0.to(DEV).T + b0.to(DEV)
    assert (c.float() - ref).abs().max().item() < 0.08


@pytest.mark.parametrize("res", [False, True])
def test_row_statistics_form_matches_the_tile_kernel(res):
    """gyre_op_linear_rowstats: same rounded outputs, and the per-row (sum, sum of squares) partials - one per wave here, one per
    N tile there - add up to the same statistics."""
    L = _lib.lib()
    M, K, N = 65536, 320, 320
    x = (randn(M, K, seed=11) * 1.3).to(torch.bfloat16).to(DEV)
    w = repack_linear(randn(N, K, seed=12) / math.sqrt(K))
    b = repack_bias(randn(N, seed=13) * 0.3)
    r = randn(M, N, seed=14).to(torch.bfloat16).to(DEV) if res else None

    def run():
        parts = L.gyre_op_linear_rowstats_parts(M, K, N, 1 if res else 0)
        assert parts > 0
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        stats = torch.full((parts, M, 2), float("nan"), dtype=torch.float32, device=DEV)
        _lib.check(L.gyre_op_linear_rowstats(st(), vp(x), M, K, vp(w), N, vp(b), vp(r), vp(y), vp(stats)))
        return y, stats
    (ya, sa), (yc, sc) = _both(run)
    assert sc.shape[0] == 4 and torch.equal(ya, yc)
    ta, tc = sa.sum(0), sc.sum(0)
    assert torch.allclose(ta, tc, rtol=1e-5, atol=1e-3)
    yf = yc.float()
    assert torch.allclose(tc[:, 0], yf.sum(1), rtol=1e-4, atol=1e-2) and torch.allclose(tc[:, 1], (yf * yf).sum(1), rtol=1e-4, atol=1e-2)
