"""The A-resident GEMM kernel (gyre_amd/csrc/kernels_gemm_ar.hip, tile config 30; -m gpu): every Linear / 1x1 projection and
GEGLU FF1 with K = 320 / 640 - the transformer blocks of the 64x64 and 32x32 UNet levels (reference call site
gyre/pipeline/unet/core.py:274 -> diffusers BasicTransformerBlock).  Single operators reach it through the C ABI once a packed-
weight scratch buffer is registered (gyre_debug_set_ar_workspace); every case is checked against the ATen fp32 op on
bf16-rounded inputs AND against the 8-wave tile kernels it replaces (same K summation order: plain / bias / residual / GEGLU
outputs must be bit-identical)."""
import math

import pytest
import torch
import torch.nn.functional as F

from gyre_amd import _lib
from gpu_util import HDT, DEV, bf16_round, randn, rel_l2, repack_bias, repack_linear, report, st, vp

pytestmark = pytest.mark.gpu
TOL = 4e-3


def to_dev_bf16(t):
    return t.to(HDT).contiguous().to(DEV)


@pytest.fixture()
def ar():
    """Registers the packed-weight scratch; yields a switch: ar(True) = planner may use config 30, ar(False) = as before."""
    L = _lib.lib()
    ws = torch.empty(5120 * 640 * 2, dtype=torch.uint8, device=DEV)

    def switch(on):
        torch.cuda.synchronize()
        L.gyre_debug_set_ar_workspace(vp(ws) if on else None, ws.numel() if on else 0)
        L.gyre_debug_gemm_ablation(0x400000 if on else 0)      # bit 22: the planner takes config 30 for the C x C projections too
    yield switch
    switch(False)


def _classes(fn):
    """Kernel classes a call launched (gyre_prof_*): proves which kernel ran."""
    _lib.prof_enable(None)
    fn()
    torch.cuda.synchronize()
    got = _lib.prof_collect()
    _lib.prof_enable([])
    return set(got)


@pytest.mark.parametrize("M,K,N,bias,res", [
    (65536, 320, 320, True, True), (65536, 320, 320, False, False), (16384, 320, 640, True, True), (16384, 320, 1920, False, False),
    (32768, 320, 960, True, False), (4096, 320, 320, True, True), (40000 + 24, 320, 320, True, True), (5000, 320, 640, False, True),
    (8192, 320, 1280, True, False),
])
def test_linear_matches_reference_and_the_tile_kernels(ar, M, K, N, bias, res):
    L = _lib.lib()
    x = bf16_round(randn(M, K, seed=9))
    w = bf16_round(randn(N, K, seed=10) / math.sqrt(K))
    b = randn(N, seed=11) if bias else None
    r = bf16_round(randn(M, N, seed=12)) if res else None
    ref = F.linear(x, w, b) + (r if res else 0)
    xd, wd = to_dev_bf16(x), repack_linear(w)
    bd, rd = (b.to(DEV) if bias else None), (to_dev_bf16(r) if res else None)
    outs = []
    for on in (True, False):
        ar(on)
        y = torch.full((M, N), float("nan"), dtype=HDT, device=DEV)
        run = lambda: _lib.check(L.gyre_op_linear(st(), vp(xd), M, K, vp(wd), N, vp(bd), vp(rd), 0, vp(y)))
        names = _classes(run)
        assert ("k_gemm_ar" in names) == on, names
        outs.append(y)
    report(f"ar linear M{M} K{K} N{N}", outs[0].float().cpu(), ref, TOL)
    assert torch.equal(outs[0], outs[1]), "same K order, same epilogue arithmetic: bit-identical to the tile kernels"


@pytest.mark.parametrize("M,K,F_", [(65536, 320, 1280), (16384, 640, 2560), (8192, 320, 1280), (4096 + 32, 640, 2560)])
def test_geglu(ar, M, K, F_):
    L = _lib.lib()
    x = bf16_round(randn(M, K, seed=13))
    w = bf16_round(randn(2 * F_, K, seed=14) / math.sqrt(K))
    b = randn(2 * F_, seed=15) * 0.5
    val, gate = F.linear(x, w, b).chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    xd, wd, bd = to_dev_bf16(x), repack_linear(w, geglu=True), repack_bias(b, geglu=True)
    outs = []
    for on in (True, False):
        ar(on)
        y = torch.full((M, F_), float("nan"), dtype=HDT, device=DEV)
        names = _classes(lambda: _lib.check(L.gyre_op_linear(st(), vp(xd), M, K, vp(wd), F_, vp(bd), None, 1, vp(y))))
        assert ("k_gemm_ar" in names) == on, names
        outs.append(y)
    report(f"ar geglu M{M} K{K} F{F_}", outs[0].float().cpu(), ref, TOL)
    assert torch.equal(outs[0], outs[1])


def _ln_inputs(M, K, seed, offset=0.3, scale=1.7):
    x = bf16_round(randn(M, K, seed=seed) * scale + offset)
    return x, randn(K, seed=seed + 1) * 0.2 + 1.0, randn(K, seed=seed + 2) * 0.2


@pytest.mark.parametrize("M,K,N,geglu", [(65536, 320, 320, 0), (65536, 320, 1280, 1), (16384, 320, 640, 0), (16384, 640, 2560, 1),
                                         (8192 + 100, 320, 320, 0)])
def test_folded_layernorm(ar, M, K, N, geglu):
    """LayerNorm folded into the GEMM (row statistics from one streaming pass, gamma in the weights, normalisation in the
    epilogue), plain and GEGLU: against the fp32 reference and against the tile kernels' folded form."""
    L = _lib.lib()
    x, g, b = _ln_inputs(M, K, 90)
    rows = 2 * N if geglu else N
    w = bf16_round(randn(rows, K, seed=93) / math.sqrt(K))
    bias_t = randn(rows, seed=94) * 0.5
    h = F.linear(F.layer_norm(x, (K,), g, b, 1e-5), w, bias_t)
    if geglu:
        val, gate = h.chunk(2, dim=-1)
        h = val * F.gelu(gate)
    xd, wd = to_dev_bf16(x), repack_linear(w, geglu=bool(geglu))
    bd = repack_bias(bias_t, geglu=True) if geglu else bias_t.to(DEV)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(rows, K, M), dtype=torch.uint8, device=DEV)
    outs = []
    for on in (True, False):
        ar(on)
        y = torch.full((M, N), float("nan"), dtype=HDT, device=DEV)
        rcs = []
        names = _classes(lambda: rcs.append(L.gyre_op_ln_linear(st(), vp(xd), M, K, vp(g.to(DEV)), vp(b.to(DEV)), 1e-5, vp(wd), N, vp(bd),
                                                                geglu, 0, None, 0, None, 0, vp(ws), ws.numel(), vp(y))))
        if not on and rcs[0] == -6:       # the tile planner has no folded form for this shape (4-wave kernel): nothing to compare with
            continue
        _lib.check(rcs[0])
        assert ("k_gemm_ar" in names) == on, names
        outs.append(y)
    report(f"ar ln_linear M{M} K{K} N{N} geglu={geglu}", outs[0].float().cpu(), h, TOL)
    if len(outs) == 2:
        assert torch.equal(outs[0], outs[1])          # the same fp32 expression per element


@pytest.mark.parametrize("M,C,res", [(65536, 320, True), (16384, 320, True), (40000 + 24, 320, False), (4096, 320, True)])
def test_row_statistics_feed_the_folded_layernorm(ar, M, C, res):
    """A C x C projection (+ residual) leaves per row the sums of its rounded outputs; the next GEMM's folded LayerNorm finishes
    mean / rstd from them."""
    L = _lib.lib()
    ar(True)
    parts = L.gyre_op_linear_rowstats_parts(M, C, C, 1 if res else 0)
    assert parts >= 1
    x = bf16_round(randn(M, C, seed=110))
    w1 = bf16_round(randn(C, C, seed=111) / math.sqrt(C))
    b1 = randn(C, seed=112) * 0.3 + 0.2
    r = bf16_round(randn(M, C, seed=113) + 0.5) if res else None
    y1 = torch.empty(M, C, dtype=HDT, device=DEV)
    stats = torch.full((parts, M, 2), float("nan"), device=DEV)
    names = _classes(lambda: _lib.check(L.gyre_op_linear_rowstats(st(), vp(to_dev_bf16(x)), M, C, vp(repack_linear(w1)), C, vp(b1.to(DEV)),
                                                                  vp(to_dev_bf16(r)) if res else None, vp(y1), vp(stats))))
    assert "k_gemm_ar" in names, names
    report(f"ar linear+rowstats M{M} C{C}", y1.float().cpu(), F.linear(x, w1, b1) + (r if res else 0), TOL)
    y1f = y1.float()
    tot = stats.sum(0)
    assert torch.allclose(tot[:, 0], y1f.sum(1), rtol=1e-4, atol=2e-3)
    assert torch.allclose(tot[:, 1], (y1f * y1f).sum(1), rtol=1e-4, atol=2e-3)
    g, b = randn(C, seed=114) * 0.2 + 1, randn(C, seed=115) * 0.2
    w2 = bf16_round(randn(C, C, seed=116) / math.sqrt(C))
    ref2 = F.linear(F.layer_norm(y1f.cpu(), (C,), g, b, 1e-5), w2)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(C, C, M), dtype=torch.uint8, device=DEV)
    out_p = torch.empty(M, C, dtype=HDT, device=DEV)
    out_s = torch.empty(M, C, dtype=HDT, device=DEV)
    args = (st(), vp(y1), M, C, vp(g.to(DEV)), vp(b.to(DEV)), 1e-5, vp(repack_linear(w2)), C, None, 0, 0, None, 0)
    _lib.check(L.gyre_op_ln_linear(*args, vp(stats), parts, vp(ws), ws.numel(), vp(out_p)))
    _lib.check(L.gyre_op_ln_linear(*args, None, 0, vp(ws), ws.numel(), vp(out_s)))
    report(f"ar ln_linear from producer statistics M{M} C{C}", out_p.float().cpu(), ref2, TOL)
    assert rel_l2(out_p.float().cpu(), out_s.float().cpu()) < 2e-3


def test_rows_are_independent_and_repeatable(ar):
    """A row's result depends on nothing but the row: the same rows inside a smaller problem (other grid, other N-range split)
    give the same bits; two runs give the same bits (counted-vmcnt ring)."""
    L = _lib.lib()
    ar(True)
    K, N = 320, 2560
    w = repack_linear(bf16_round(randn(N, K, seed=3) / math.sqrt(K)), geglu=True)
    b = repack_bias(randn(N, seed=4) * 0.3, geglu=True)
    x = to_dev_bf16(randn(65536, K, seed=5))
    ys = []
    for M in (65536, 65536, 8192, 4096 + 64):
        y = torch.full((M, N // 2), float("nan"), dtype=HDT, device=DEV)
        _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), N // 2, vp(b), None, 1, vp(y)))
        ys.append(y)
    assert torch.equal(ys[0], ys[1])
    assert torch.equal(ys[0][:8192], ys[2]) and torch.equal(ys[0][:4096 + 64], ys[3])


@pytest.mark.parametrize("B,tokens,ln", [(16, 4096, True), (2, 4096, True), (4, 4096, False), (5, 1056, True), (9, 1024, False)])
def test_fused_qkv_with_transposed_v(ar, B, tokens, ln):
    """Q | K | V of a self-attention as one launch of the A-resident kernel: Q | K row-major, the V blocks transposed through
    the per-wave LDS patch into V^T[b][c][token]; with and without the folded LayerNorm, one and several N-range splits."""
    L = _lib.lib()
    C, M = 320, B * tokens
    x, g, b = _ln_inputs(M, C, 100)
    w = bf16_round(randn(3 * C, C, seed=103) / math.sqrt(C))
    ref = F.linear(F.layer_norm(x, (C,), g, b, 1e-5) if ln else x, w)
    xd, wd = to_dev_bf16(x), repack_linear(w)
    outs = []
    for on in (True, False):
        ar(on)
        qk = torch.full((M, 2 * C), float("nan"), dtype=HDT, device=DEV)
        vt = torch.full((B, C, tokens), float("nan"), dtype=HDT, device=DEV)
        rcs = []
        if ln:
            ws = torch.empty(L.gyre_op_ln_linear_workspace(3 * C, C, M), dtype=torch.uint8, device=DEV)
            call = lambda: rcs.append(L.gyre_op_ln_linear(st(), vp(xd), M, C, vp(g.to(DEV)), vp(b.to(DEV)), 1e-5, vp(wd), 3 * C, None, 0,
                                                          tokens, vp(vt), tokens, None, 0, vp(ws), ws.numel(), vp(qk)))
        else:
            call = lambda: rcs.append(L.gyre_op_qkv(st(), vp(xd), M, C, vp(wd), tokens, vp(qk), vp(vt), tokens))
        names = _classes(call)
        if not on and rcs[0] == -6:
            continue
        _lib.check(rcs[0])
        assert ("k_gemm_ar" in names) == on, names
        report(f"ar qkv QK part B{B} T{tokens} ln={ln} ar={on}", qk.float().cpu(), ref[:, :2 * C], TOL)
        report(f"ar qkv V^T part B{B} T{tokens} ln={ln} ar={on}", vt.float().cpu(), ref[:, 2 * C:].reshape(B, tokens, C).permute(0, 2, 1), TOL)
        outs.append((qk, vt))
    if len(outs) == 2:
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
