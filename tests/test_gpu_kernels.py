"""Kernel-level parity (-m gpu): every hand-written HIP kernel, called through the C ABI,
against the same ATen fp32 op the oracle uses, on bf16-rounded inputs.

Tolerances (bf16 storage, fp32 accumulate): outputs are rounded to bf16 once
(relative 2^-9 per element => rel-L2 ~1.5e-3); attention additionally quantises P to bf16.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from gyre_amd import _lib
from gpu_util import HDT, DEV, bf16_round, randn, rel_l2, repack_bias, repack_conv, repack_linear, report, st, vp

pytestmark = pytest.mark.gpu

TOL = 4e-3
TOL_ATTN = 1e-2


def to_dev_bf16(t):
    return t.to(HDT).contiguous().to(DEV)


def nhwc(t):  # NCHW f32 -> NHWC
    return t.permute(0, 2, 3, 1).contiguous()


def test_nchw_to_nhwc_pad():
    L = _lib.lib()
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        x = randn(2, 9, 12, 20, seed=1).to(dt)
        xd = x.to(DEV)
        y = torch.full((2, 12 * 20, 16), 7.0, dtype=HDT, device=DEV)
        _lib.check(L.gyre_op_nchw_to_nhwc(st(), vp(xd), _lib.dtype_code(xd), 2, 9, 240, 16, vp(y)))
        ref = torch.zeros(2, 240, 16)
        ref[:, :, :9] = bf16_round(x.float()).reshape(2, 9, 240).permute(0, 2, 1)
        assert torch.equal(y.float().cpu(), ref)


@pytest.mark.parametrize("B,H,W,C,C1,silu,eps", [
    (2, 64, 64, 320, 0, 1, 1e-5), (2, 32, 32, 640, 0, 0, 1e-6), (2, 16, 16, 1280, 0, 1, 1e-5),
    (2, 8, 8, 2560, 1280, 1, 1e-5), (2, 32, 32, 960, 640, 1, 1e-5), (1, 128, 128, 128, 0, 1, 1e-6),
    (3, 24, 40, 64, 0, 1, 1e-5), (1, 8, 8, 1920, 1280, 1, 1e-5),
    (2, 16, 16, 1280, 0, 0, 1e-6), (2, 16, 16, 2560, 1280, 1, 1e-5), (2, 16, 16, 1920, 1280, 1, 1e-5),
    (2, 32, 32, 640, 0, 1, 1e-5), (4, 8, 8, 1280, 0, 1, 1e-5), (2, 8, 8, 64, 0, 1, 1e-5),
])
def test_groupnorm(B, H, W, C, C1, silu, eps):
    L = _lib.lib()
    x = bf16_round(randn(B, C, H, W, seed=2) * 1.5 + 0.7)
    gamma, beta = randn(C, seed=3) * 0.2 + 1.0, randn(C, seed=4) * 0.3
    ref = F.group_norm(x, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    xn = nhwc(x)
    if C1:
        a, b = to_dev_bf16(xn[..., :C1]), to_dev_bf16(xn[..., C1:])
    else:
        a, b = to_dev_bf16(xn), None
    y = torch.empty(B, H, W, C, dtype=HDT, device=DEV)
    wsb = L.gyre_op_groupnorm_workspace(B, H * W, C, 32)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    _lib.check(L.gyre_op_groupnorm(st(), vp(a), vp(b), C1, B, H * W, C, 32, vp(gamma.to(DEV)), vp(beta.to(DEV)), eps,
                                   silu, vp(ws), wsb, vp(y)))
    report(f"groupnorm {B}x{H}x{W}x{C} C1={C1}", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)


def test_groupnorm_batch_independent():
    """bit-identical result for a sample whether normalised alone or inside a batch."""
    L = _lib.lib()
    x = to_dev_bf16(nhwc(randn(3, 320, 32, 32, seed=5)))
    g, b = torch.ones(320, device=DEV), torch.zeros(320, device=DEV)

    def run(t):
        B = t.shape[0]
        y = torch.empty_like(t)
        wsb = L.gyre_op_groupnorm_workspace(B, 1024, 320, 32)
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        _lib.check(L.gyre_op_groupnorm(st(), vp(t), None, 0, B, 1024, 320, 32, vp(g), vp(b), 1e-5, 1, vp(ws), wsb, vp(y)))
        return y
    full = run(x)
    one = run(x[1:2].contiguous())
    assert torch.equal(full[1:2], one)


@pytest.mark.parametrize("M,C", [(1000, 320), (513, 640), (300, 1280), (64, 64), (77, 2048)])
def test_layernorm(M, C):
    L = _lib.lib()
    x = bf16_round(randn(M, C, seed=6) * 2 + 0.3)
    g, b = randn(C, seed=7) * 0.2 + 1, randn(C, seed=8) * 0.2
    ref = F.layer_norm(x, (C,), g, b, 1e-5)
    y = torch.empty(M, C, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_layernorm(st(), vp(to_dev_bf16(x)), M, C, vp(g.to(DEV)), vp(b.to(DEV)), 1e-5, vp(y)))
    report(f"layernorm {M}x{C}", y.float().cpu(), ref, TOL)


@pytest.mark.parametrize("M,K,N,bias,res", [
    (4096, 320, 320, True, True), (1232, 768, 320, False, False), (300, 320, 1280, True, False),
    (2048, 1280, 1280, True, True), (8192, 320, 640, False, False), (65, 64, 32, True, True),
    (32768, 320, 320, True, True), (16384, 1280, 640, True, False), (16, 320, 1280, True, False),
    (1024, 5120, 1280, True, True), (5000, 8, 8, True, False),
])
def test_linear(M, K, N, bias, res):
    L = _lib.lib()
    x = bf16_round(randn(M, K, seed=9))
    w = bf16_round(randn(N, K, seed=10) / math.sqrt(K))
    b = randn(N, seed=11) if bias else None
    r = bf16_round(randn(M, N, seed=12)) if res else None
    ref = F.linear(x, w, b)
    if res:
        ref = ref + r
    y = torch.empty(M, N, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_linear(st(), vp(to_dev_bf16(x)), M, K, vp(repack_linear(w)), N,
                                vp(b.to(DEV)) if bias else None, vp(to_dev_bf16(r)) if res else None, 0, vp(y)))
    report(f"linear M{M} K{K} N{N}", y.float().cpu(), ref, TOL)


@pytest.mark.parametrize("M,K,F_", [(1024, 320, 1280), (4096, 640, 2560), (200, 64, 256), (16384, 320, 1280)])
def test_linear_geglu(M, K, F_):
    L = _lib.lib()
    x = bf16_round(randn(M, K, seed=13))
    w = bf16_round(randn(2 * F_, K, seed=14) / math.sqrt(K))
    b = randn(2 * F_, seed=15) * 0.5
    h = F.linear(x, w, b)
    val, gate = h.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    y = torch.empty(M, F_, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_linear(st(), vp(to_dev_bf16(x)), M, K, vp(repack_linear(w, geglu=True)), F_,
                                vp(repack_bias(b, geglu=True)), None, 1, vp(y)))
    report(f"geglu M{M} K{K} F{F_}", y.float().cpu(), ref, TOL)


@pytest.mark.parametrize("B,T,K,N,bias", [(2, 1024, 320, 320, False), (2, 77, 768, 640, False), (3, 64, 1280, 1280, True),
                                          (1, 4096, 512, 512, True), (2, 4096, 320, 320, False)])
def test_linear_transposed(B, T, K, N, bias):
    L = _lib.lib()
    ldt = (T + 7) // 8 * 8
    x = bf16_round(randn(B * T, K, seed=16))
    w = bf16_round(randn(N, K, seed=17) / math.sqrt(K))
    b = randn(N, seed=18) if bias else None
    ref = F.linear(x, w, b).reshape(B, T, N).permute(0, 2, 1)  # [B][N][T]
    y = torch.zeros(B, N, ldt, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_linear_t(st(), vp(to_dev_bf16(x)), B * T, K, vp(repack_linear(w)), N,
                                  vp(b.to(DEV)) if bias else None, T, ldt, vp(y)))
    report(f"linear_t B{B} T{T} K{K} N{N}", y.float().cpu()[:, :, :T], ref, TOL)
    assert float(y[:, :, T:].float().abs().max().cpu() if ldt > T else 0.0) == 0.0  # pad columns untouched


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups,asym,res", [
    (2, 32, 32, 320, 320, 1, 0, 0, True), (2, 32, 32, 320, 640, 1, 0, 0, False), (2, 32, 32, 320, 320, 2, 0, 0, False),
    (2, 16, 16, 640, 640, 1, 1, 0, False), (1, 32, 32, 128, 128, 2, 0, 1, False), (2, 24, 40, 64, 96, 1, 0, 0, True),
    (2, 64, 64, 8, 320, 1, 0, 0, False), (2, 16, 16, 16, 64, 1, 0, 0, False), (1, 8, 8, 2560, 1280, 1, 0, 0, False),
    (1, 64, 64, 960, 320, 1, 0, 0, True), (3, 8, 8, 1280, 1280, 1, 0, 0, True), (1, 20, 12, 64, 64, 2, 0, 0, False),
    (1, 64, 64, 128, 8, 1, 0, 0, False),
])
def test_conv3x3(B, H, W, Cin, Cout, stride, ups, asym, res):
    L = _lib.lib()
    x = bf16_round(randn(B, Cin, H, W, seed=19))
    w = bf16_round(randn(Cout, Cin, 3, 3, seed=20) / math.sqrt(9 * Cin))
    b = randn(Cout, seed=21)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    if asym:
        ref = F.conv2d(F.pad(xi, (0, 1, 0, 1)), w, b, stride=stride, padding=0)
    else:
        ref = F.conv2d(xi, w, b, stride=stride, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    r = bf16_round(randn(B, Cout, Ho, Wo, seed=22)) if res else None
    if res:
        ref = ref + r
    y = torch.empty(B, Ho, Wo, Cout, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_conv3x3(st(), vp(to_dev_bf16(nhwc(x))), B, H, W, Cin, vp(repack_conv(w)), Cout, vp(b.to(DEV)),
                                 vp(to_dev_bf16(nhwc(r))) if res else None, stride, ups, asym, vp(y)))
    report(f"conv3x3 {B}x{H}x{W} {Cin}->{Cout} s{stride} ups{ups} asym{asym}", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)


@pytest.mark.parametrize("B,H,W,Cin,Cout,dt", [
    (2, 64, 64, 320, 4, torch.float32), (16, 64, 64, 320, 4, torch.bfloat16), (1, 96, 96, 320, 4, torch.float16),
    (3, 13, 21, 320, 4, torch.float32), (1, 8, 8, 64, 1, torch.float32), (2, 40, 24, 128, 3, torch.float32),
    (1, 64, 64, 192, 9, torch.float32), (2, 16, 16, 32, 4, torch.float32),
])
def test_conv_out_nchw(B, H, W, Cin, Cout, dt):
    """The models' output convolution (UNet conv_out 320 -> 4, VAE decoder conv_out 128 -> 3; reference call sites
    unet/core.py:274, unified_pipeline.py:1531) on its dedicated kernel (kernels_conv_out.hip; round 5): against F.conv2d, against
    the tile kernels it replaces, at image sizes that are no multiple of its 8 x 8 tile, in the three boundary dtypes, and (Cin = 32)
    the shapes that still take the tile kernels.  Bit-reproducible, and a sample's result does not depend on the batch around it."""
    L = _lib.lib()
    x = bf16_round(randn(B, Cin, H, W, seed=26))
    w = bf16_round(randn(Cout, Cin, 3, 3, seed=27) / math.sqrt(9 * Cin))
    b = randn(Cout, seed=28)
    ref = F.conv2d(x, w, b, padding=1)
    xd, wd, bd = to_dev_bf16(nhwc(x)), repack_conv(w), b.to(DEV)
    code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dt]
    tol = TOL if dt == torch.float32 else 2 * TOL
    y = torch.full((B, Cout, H, W), float("nan"), dtype=dt, device=DEV)
    _lib.check(L.gyre_op_conv3x3_nchw(st(), vp(xd), B, H, W, Cin, vp(wd), Cout, vp(bd), vp(y), code, 0))
    report(f"conv_out {B}x{H}x{W} {Cin}->{Cout} {dt}", y.float().cpu(), ref, tol)
    y2 = torch.full_like(y, float("nan"))
    _lib.check(L.gyre_op_conv3x3_nchw(st(), vp(xd), B, H, W, Cin, vp(wd), Cout, vp(bd), vp(y2), code, 0))
    assert torch.equal(y, y2)
    yt = torch.full_like(y, float("nan"))
    _lib.check(L.gyre_op_conv3x3_nchw(st(), vp(xd), B, H, W, Cin, vp(wd), Cout, vp(bd), vp(yt), code, 1))     # the tile kernels
    report(f"conv_out vs tile kernels {B}x{H}x{W} {Cin}->{Cout}", y.float().cpu(), yt.float().cpu(), tol)
    if Cin % 64 == 0:                                # same K order (64-channel chunks, taps inside, two MFMA steps per tap): same bits
        assert torch.equal(y, yt)
    if B > 1:                                        # the last sample alone: same bits as inside the batch
        y1 = torch.full((1, Cout, H, W), float("nan"), dtype=dt, device=DEV)
        _lib.check(L.gyre_op_conv3x3_nchw(st(), vp(xd[B - 1:].contiguous()), 1, H, W, Cin, vp(wd), Cout, vp(bd), vp(y1), code, 0))
        assert torch.equal(y1[0], y[B - 1])


def test_conv3x3_padded_cin():
    """Cin=4 latents are zero-padded to 8 channels at the NCHW boundary; weights repacked with the same pad."""
    L = _lib.lib()
    B, H, W, Cin, Cout = 2, 32, 32, 4, 320
    x = bf16_round(randn(B, Cin, H, W, seed=23))
    w = bf16_round(randn(Cout, Cin, 3, 3, seed=24) / 6)
    b = randn(Cout, seed=25)
    ref = F.conv2d(x, w, b, padding=1)
    xd = x.to(DEV)
    xp = torch.empty(B, H * W, 8, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_nchw_to_nhwc(st(), vp(xd), 0, B, Cin, H * W, 8, vp(xp)))
    y = torch.empty(B, H, W, Cout, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_conv3x3(st(), vp(xp), B, H, W, 8, vp(repack_conv(w, 8)), Cout, vp(b.to(DEV)), None, 1, 0, 0, vp(y)))
    report("conv3x3 cin4->pad8", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)


def attn_ref(q, k, v, heads):
    B, Nq, C = q.shape
    d = C // heads
    sp = lambda t: t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)
    s = (sp(q) @ sp(k).transpose(-1, -2)) * d ** -0.5
    o = s.softmax(-1) @ sp(v)
    return o.permute(0, 2, 1, 3).reshape(B, Nq, C)


@pytest.mark.parametrize("B,heads,Nq,Nk,D", [
    (1, 2, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (2, 8, 256, 256, 160), (2, 8, 64, 64, 160),
    (2, 8, 1024, 77, 40), (1, 8, 256, 231, 160), (2, 4, 96, 77, 32), (1, 1, 1024, 1024, 512), (2, 2, 100, 50, 16),
    (1, 10, 1024, 1024, 64), (1, 2, 1000, 333, 128),
])
def test_attention(B, heads, Nq, Nk, D):
    L = _lib.lib()
    C_ = heads * D
    q = bf16_round(randn(B, Nq, C_, seed=26))
    k = bf16_round(randn(B, Nk, C_, seed=27))
    v = bf16_round(randn(B, Nk, C_, seed=28))
    ref = attn_ref(q, k, v, heads)
    ldvt = (Nk + 7) // 8 * 8
    vt = torch.full((B, C_, ldvt), float("nan"), dtype=HDT, device=DEV)  # pad = NaN on purpose
    vt[:, :, :Nk] = v.permute(0, 2, 1).to(HDT).to(DEV)
    o = torch.empty(B, Nq, C_, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_attention(st(), vp(to_dev_bf16(q)), C_, vp(to_dev_bf16(k)), C_, vp(vt), ldvt, B, heads, Nq, Nk, D,
                                   vp(o), C_))
    report(f"attention B{B} h{heads} Nq{Nq} Nk{Nk} D{D}", o.float().cpu(), ref, TOL_ATTN)


def test_attention_peaked_softmax():
    """Forces large score ranges / running-max jumps across KV tiles (online-softmax rescale path)."""
    L = _lib.lib()
    B, heads, N, D = 1, 2, 512, 40
    C_ = heads * D
    q = bf16_round(randn(B, N, C_, seed=29) * 4)
    k = bf16_round(randn(B, N, C_, seed=30) * 4)
    k[:, 300] = q[:, 7] * 3  # one key aligned with one query, located in a late tile
    k = bf16_round(k)
    v = bf16_round(randn(B, N, C_, seed=31))
    ref = attn_ref(q, k, v, heads)
    vt = v.permute(0, 2, 1).to(HDT).contiguous().to(DEV)
    o = torch.empty(B, N, C_, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_attention(st(), vp(to_dev_bf16(q)), C_, vp(to_dev_bf16(k)), C_, vp(vt), N, B, heads, N, N, D,
                                   vp(o), C_))
    report("attention peaked", o.float().cpu(), ref, 2e-2)


def test_error_paths():
    L = _lib.lib()
    x = torch.zeros(64, 12, dtype=HDT, device=DEV)
    rc = L.gyre_op_linear(st(), vp(x), 64, 12, vp(x), 8, None, None, 0, vp(x))
    assert rc == -1 and b"multiple of 8" in L.gyre_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    rc = L.gyre_op_attention(st(), vp(x), 8, vp(x), 8, vp(x), 8, 1, 1, 8, 8, 12, vp(x), 8)
    assert rc == -1
    rc = L.gyre_op_attention(st(), vp(x), 48, vp(x), 48, vp(x), 8, 1, 1, 8, 8, 48, vp(x), 48)
    assert rc == -6
    with pytest.raises(NotImplementedError):
        _lib.check(rc)


# ---- every GEMM tile configuration, forced, on shapes with M / N / K tails -------------------------
@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("M,K,N,bias,res", [(1000, 320, 640, True, True), (4096 + 37, 200, 1280, True, False),
                                            (513, 1280, 320 * 4, False, True)])
def test_linear_forced_tile_config(cfg, M, K, N, bias, res):
    L = _lib.lib()
    x = bf16_round(randn(M, K, seed=40))
    w = bf16_round(randn(N, K, seed=41) / math.sqrt(K))
    b = randn(N, seed=42) if bias else None
    r = bf16_round(randn(M, N, seed=43)) if res else None
    ref = F.linear(x, w, b)
    if res:
        ref = ref + r
    y = torch.empty(M, N, dtype=HDT, device=DEV)
    old = L.gyre_debug_force_gemm_cfg(cfg)
    try:
        _lib.check(L.gyre_op_linear(st(), vp(to_dev_bf16(x)), M, K, vp(repack_linear(w)), N,
                                    vp(b.to(DEV)) if bias else None, vp(to_dev_bf16(r)) if res else None, 0, vp(y)))
    finally:
        L.gyre_debug_force_gemm_cfg(old)
    report(f"linear cfg{cfg} M{M} K{K} N{N}", y.float().cpu(), ref, TOL)


@pytest.mark.parametrize("cfg", [1, 2, 3, 6, 7])
def test_geglu_forced_tile_config(cfg):
    L = _lib.lib()
    M, K, F_ = 700, 320, 1280
    x = bf16_round(randn(M, K, seed=44))
    w = bf16_round(randn(2 * F_, K, seed=45) / math.sqrt(K))
    b = randn(2 * F_, seed=46) * 0.5
    val, gate = F.linear(x, w, b).chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    y = torch.empty(M, F_, dtype=HDT, device=DEV)
    old = L.gyre_debug_force_gemm_cfg(cfg)
    try:
        _lib.check(L.gyre_op_linear(st(), vp(to_dev_bf16(x)), M, K, vp(repack_linear(w, geglu=True)), F_,
                                    vp(repack_bias(b, geglu=True)), None, 1, vp(y)))
    finally:
        L.gyre_debug_force_gemm_cfg(old)
    report(f"geglu cfg{cfg}", y.float().cpu(), ref, TOL)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups,asym,res", [
    (2, 24, 20, 320, 640, 1, 0, 0, True), (1, 16, 16, 72, 1280, 1, 1, 0, False), (2, 18, 18, 128, 1280, 2, 0, 1, False),
])
def test_conv_forced_tile_config(cfg, B, H, W, Cin, Cout, stride, ups, asym, res):
    L = _lib.lib()
    x = bf16_round(randn(B, Cin, H, W, seed=47))
    w = bf16_round(randn(Cout, Cin, 3, 3, seed=48) / math.sqrt(9 * Cin))
    b = randn(Cout, seed=49)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(F.pad(xi, (0, 1, 0, 1)), w, b, stride=stride, padding=0) if asym else F.conv2d(xi, w, b, stride=stride, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    r = bf16_round(randn(B, Cout, Ho, Wo, seed=50)) if res else None
    if res:
        ref = ref + r
    y = torch.empty(B, Ho, Wo, Cout, dtype=HDT, device=DEV)
    old = L.gyre_debug_force_gemm_cfg(cfg)
    try:
        _lib.check(L.gyre_op_conv3x3(st(), vp(to_dev_bf16(nhwc(x))), B, H, W, Cin, vp(repack_conv(w)), Cout, vp(b.to(DEV)),
                                     vp(to_dev_bf16(nhwc(r))) if res else None, stride, ups, asym, vp(y)))
    finally:
        L.gyre_debug_force_gemm_cfg(old)
    report(f"conv cfg{cfg} {B}x{H}x{W} {Cin}->{Cout}", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)


@pytest.mark.parametrize("B,H,W,Cin,Cout,res", [(16, 8, 8, 1280, 1280, True), (4, 16, 16, 640, 1280, False),
                                                (16, 8, 8, 2560, 1280, True), (2, 8, 8, 512, 512, False)])
def test_conv_split_k(B, H, W, Cin, Cout, res):
    """Few output tiles + long K: the planner cuts K into slices (fp32 slabs + deterministic reduce)."""
    L = _lib.lib()
    x = bf16_round(randn(B, Cin, H, W, seed=51))
    w = bf16_round(randn(Cout, Cin, 3, 3, seed=52) / math.sqrt(9 * Cin))
    b = randn(Cout, seed=53)
    ref = F.conv2d(x, w, b, padding=1)
    r = bf16_round(randn(B, Cout, H, W, seed=54)) if res else None
    if res:
        ref = ref + r
    y = torch.empty(B, H, W, Cout, dtype=HDT, device=DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    L.gyre_debug_set_splitk_workspace(vp(ws), ws.numel())
    try:
        args = (st(), vp(to_dev_bf16(nhwc(x))), B, H, W, Cin, vp(repack_conv(w)), Cout, vp(b.to(DEV)),
                vp(to_dev_bf16(nhwc(r))) if res else None, 1, 0, 0, vp(y))
        _lib.check(L.gyre_op_conv3x3(*args))
        y1 = y.clone()
        _lib.check(L.gyre_op_conv3x3(*args))
        assert torch.equal(y, y1)  # deterministic reduction order
    finally:
        L.gyre_debug_set_splitk_workspace(None, 0)
    report(f"conv split-K {B}x{H}x{W} {Cin}->{Cout}", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)


@pytest.mark.parametrize("variant", [1, 2, 4])
@pytest.mark.parametrize("B,heads,Nq,Nk,D", [(1, 2, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (2, 8, 256, 256, 160),
                                              (2, 8, 1024, 77, 40), (1, 8, 256, 231, 160), (2, 4, 96, 77, 32),
                                              (2, 2, 100, 50, 16), (1, 10, 1024, 1000, 64), (1, 2, 1000, 333, 128)])
def test_attention_variants(variant, B, heads, Nq, Nk, D):
    """register-staged (1) and LDS-DMA double-buffered (2: 32 q rows / wave, 4: 64) kernels, incl. key tails."""
    L = _lib.lib()
    C_ = heads * D
    q = bf16_round(randn(B, Nq, C_, seed=60))
    k = bf16_round(randn(B, Nk, C_, seed=61))
    v = bf16_round(randn(B, Nk, C_, seed=62))
    ref = attn_ref(q, k, v, heads)
    ldvt = (Nk + 7) // 8 * 8
    vt = torch.zeros((B, C_, ldvt), dtype=HDT, device=DEV)  # pad columns must be finite (zero) for v2
    vt[:, :, :Nk] = v.permute(0, 2, 1).to(HDT).to(DEV)
    o = torch.empty(B, Nq, C_, dtype=HDT, device=DEV)
    old = L.gyre_debug_force_attn_variant(variant)
    try:
        _lib.check(L.gyre_op_attention(st(), vp(to_dev_bf16(q)), C_, vp(to_dev_bf16(k)), C_, vp(vt), ldvt, B, heads, Nq, Nk,
                                       D, vp(o), C_))
    finally:
        L.gyre_debug_force_attn_variant(old)
    report(f"attention v{variant} B{B} h{heads} Nq{Nq} Nk{Nk} D{D}", o.float().cpu(), ref, TOL_ATTN)


def test_all_tile_configs_sum_in_the_same_order():
    """Every production tile config (4-wave 128x128 / 256x64 / 64x64, 8-wave 256x320 / 128x320 / 256x256 / 128x256 / 128x160)
    gives BIT-identical output for the same problem: the planner may pick by problem size without changing results."""
    L = _lib.lib()
    # 3x3 conv, uniform taps, dual-source-free; Cout multiple of 320 and 256 so that all configs are legal
    B, H, W, Cin, Cout = 2, 20, 24, 192, 1280
    x = to_dev_bf16(nhwc(bf16_round(randn(B, Cin, H, W, seed=60))))
    w = repack_conv(bf16_round(randn(Cout, Cin, 3, 3, seed=61) / math.sqrt(9 * Cin)))
    b = randn(Cout, seed=62).to(DEV)
    outs = {}
    for cfg in (1, 2, 3, 4, 5, 6, 7, 8):
        y = torch.empty(B, H, W, Cout, dtype=HDT, device=DEV)
        old = L.gyre_debug_force_gemm_cfg(cfg)
        try:
            _lib.check(L.gyre_op_conv3x3(st(), vp(x), B, H, W, Cin, vp(w), Cout, vp(b), None, 1, 0, 0, vp(y)))
        finally:
            L.gyre_debug_force_gemm_cfg(old)
        outs[cfg] = y
    for cfg, y in outs.items():
        assert torch.equal(y, outs[1]), f"conv: config {cfg} differs bitwise from config 1"
    # linear
    M, K, N = 777, 640, 1280
    xl = to_dev_bf16(bf16_round(randn(M, K, seed=63)))
    wl = repack_linear(bf16_round(randn(N, K, seed=64) / math.sqrt(K)))
    outs = {}
    for cfg in (1, 2, 3, 4, 5, 6, 7, 8):
        y = torch.empty(M, N, dtype=HDT, device=DEV)
        old = L.gyre_debug_force_gemm_cfg(cfg)
        try:
            _lib.check(L.gyre_op_linear(st(), vp(xl), M, K, vp(wl), N, None, None, 0, vp(y)))
        finally:
            L.gyre_debug_force_gemm_cfg(old)
        outs[cfg] = y
    for cfg, y in outs.items():
        assert torch.equal(y, outs[1]), f"linear: config {cfg} differs bitwise from config 1"


@pytest.mark.parametrize("M,K,N,res", [(777, 640, 1280, 1), (4096, 1280, 1280, 1), (300, 320, 320, 0), (1024, 2560, 640, 1), (2048, 5120, 1280, 0),
                                       (130, 1280, 960, 0)])
def test_linear_ring_loop_and_blocked_weights_are_bit_identical(M, K, N, res):
    """Round 5: the nst-deep LDS ring of the 8-wave kernels' linear K loop (counted vmcnt, 2 - 4 stages by tile and grid size;
    tuning bit 24 = the two-stage loop it replaces, bit 25 = deep ring also where two workgroups per CU would fit) and the BLOCKED
    weight copy of the LDS-DMA kernels (gyre_debug_set_wblk_workspace; K > 1024) change the order of nothing: same bits for every
    tile config, with and without split K."""
    L = _lib.lib()
    x = to_dev_bf16(bf16_round(randn(M, K, seed=63)))
    w = repack_linear(bf16_round(randn(N, K, seed=64) / math.sqrt(K)))
    b = repack_bias(randn(N, seed=65))
    r = to_dev_bf16(bf16_round(randn(M, N, seed=66))) if res else None
    wsk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    blk = torch.empty(N * K * 2, dtype=torch.uint8, device=DEV)
    L.gyre_debug_set_splitk_workspace(vp(wsk), wsk.numel())
    refs = {}
    try:
        for cfg in (4, 5, 6, 7, 8, 24, 32, 8 | (2 << 8), 5 | (2 << 8), 24 | (2 << 8)):
            if (cfg & 0xff) in (4, 5, 8, 24) and N % 320 and not ((cfg & 0xff) == 8 and N % 160 == 0): continue
            if (cfg & 0xff) in (6, 7) and N % 256: continue
            if (cfg & 0xff) == 24 and K < 2048: continue
            if (cfg & 0xff) == 32 and (N % 64 or cfg >> 8): continue
            if (cfg >> 8) and K < 2048: continue
            for bits, blocked in ((0x1000000, 0), (0, 0), (0x2000000, 0), (0, 1)):
                y = torch.full((M, N), float("nan"), dtype=HDT, device=DEV)
                L.gyre_debug_set_wblk_workspace(vp(blk) if blocked else None, blk.numel() if blocked else 0)
                oc, ob = L.gyre_debug_force_gemm_cfg(cfg), L.gyre_debug_gemm_ablation(bits)
                try:
                    _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), vp(r), 0, vp(y)))
                finally:
                    L.gyre_debug_force_gemm_cfg(oc); L.gyre_debug_gemm_ablation(ob)
                key = cfg >> 8                       # (a split-K factor has its own summation order: compared within itself)
                if key not in refs:
                    refs[key] = y
                    report(f"linear ring M{M} K{K} N{N} splits {max(key, 1)}", y.float().cpu(), F.linear(x.float().cpu(), bf16_round(randn(N, K, seed=64) / math.sqrt(K)), randn(N, seed=65)) + (r.float().cpu() if res else 0), TOL)
                assert torch.equal(y, refs[key]), f"config {cfg:#x} bits {bits:#x} blocked {blocked} differs"
    finally:
        L.gyre_debug_set_wblk_workspace(None, 0)
        L.gyre_debug_set_splitk_workspace(None, 0)


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 24, 192, 640), (1, 16, 1280, 1280), (2, 8, 640, 1280)])
def test_conv_blocked_weights_are_bit_identical(B, H, Cin, Cout):
    """3x3 convs read the blocked weight copy too (every conv has K = 9 Cin > 1024): same bits as from the row-major weights for
    the 8-wave tiles, the pipelined 32x32x16 tile and their split-K forms.  Late round 5: the 8-wave kernel's conv K loop runs on
    the nst-deep LDS ring as well (tuning bit 27 = the two-stage loop it replaces): same bits again, also against the fp32 conv."""
    L = _lib.lib()
    x = to_dev_bf16(nhwc(bf16_round(randn(B, Cin, H, H, seed=60))))
    w = repack_conv(bf16_round(randn(Cout, Cin, 3, 3, seed=61) / math.sqrt(9 * Cin)))
    b = randn(Cout, seed=62).to(DEV)
    wsk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    blk = torch.empty(Cout * 9 * Cin * 2, dtype=torch.uint8, device=DEV)
    L.gyre_debug_set_splitk_workspace(vp(wsk), wsk.numel())
    refs = {}
    try:
        for cfg in (4, 5, 8, 24, 8 | (2 << 8), 8 | (4 << 8), 5 | (2 << 8), 24 | (2 << 8)):
            for blocked, bits in ((0, 0), (1, 0), (0, 0x8000000), (1, 0x8000000)):
                if bits and (cfg & 0xff) == 24: continue          # (the pipelined tile has its own loop)
                y = torch.full((B, H, H, Cout), float("nan"), dtype=HDT, device=DEV)
                L.gyre_debug_set_wblk_workspace(vp(blk) if blocked else None, blk.numel() if blocked else 0)
                oc, ob = L.gyre_debug_force_gemm_cfg(cfg), L.gyre_debug_gemm_ablation(bits)
                try:
                    _lib.check(L.gyre_op_conv3x3(st(), vp(x), B, H, H, Cin, vp(w), Cout, vp(b), None, 1, 0, 0, vp(y)))
                finally:
                    L.gyre_debug_force_gemm_cfg(oc); L.gyre_debug_gemm_ablation(ob)
                if (cfg >> 8) not in refs:
                    refs[cfg >> 8] = y
                    ref = F.conv2d(bf16_round(randn(B, Cin, H, H, seed=60)), bf16_round(randn(Cout, Cin, 3, 3, seed=61) / math.sqrt(9 * Cin)),
                                   randn(Cout, seed=62), padding=1)
                    report(f"conv ring B{B} H{H} {Cin}->{Cout} splits {max(cfg >> 8, 1)}", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)
                assert torch.equal(y, refs[cfg >> 8]), f"config {cfg:#x} blocked {blocked} bits {bits:#x} differs"
    finally:
        L.gyre_debug_set_wblk_workspace(None, 0)
        L.gyre_debug_set_splitk_workspace(None, 0)


def test_groupnorm_statistics_do_not_depend_on_batch_size():
    """The per-sample chunking of the two-pass statistics is a function of HW only."""
    L = _lib.lib()
    for HW, Cc in ((4096, 320), (1024, 640), (9216, 320), (65536, 128)):
        B = 5
        x = (torch.randn(B, HW, Cc, device=DEV) * 2 + 0.5).to(HDT)
        gam, bet = torch.randn(Cc, device=DEV), torch.randn(Cc, device=DEV)

        def run(xs):
            n = xs.shape[0]
            y = torch.empty_like(xs)
            wsb = L.gyre_op_groupnorm_workspace(n, HW, Cc, 32)
            ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
            _lib.check(L.gyre_op_groupnorm(st(), vp(xs), None, 0, n, HW, Cc, 32, vp(gam), vp(bet), 1e-5, 1, vp(ws), wsb, vp(y)))
            return y
        full = run(x)
        assert torch.equal(run(x[3:4].contiguous()), full[3:4])
        assert torch.equal(run(x[1:4].contiguous()), full[1:4])


@pytest.mark.parametrize("B,heads,Nq,Nk,D", [(2, 8, 1024, 1024, 40), (1, 4, 300, 77, 40), (2, 2, 200, 333, 16), (1, 4, 130, 64, 32),
                                             (1, 5, 256, 1000, 64), (2, 8, 256, 256, 80), (1, 8, 64, 154, 160), (1, 8, 100, 7, 40),
                                             (1, 1, 70, 70, 512)])
@pytest.mark.parametrize("variant", [0, 3, 5])
def test_attention_prescaled_k(variant, B, heads, Nq, Nk, D):
    """gyre_op_attention_ex(k_prescaled=1): K carries log2(e)/sqrt(D) (as the UNet's scaled to_k weights produce it,
    one bf16 rounding of the fp32 product) -> folded-softmax kernel for D in {16,32,40,64,160}, plain kernels with
    unit scale otherwise.  Reference: fp32 softmax attention on the unrounded K."""
    L = _lib.lib()
    C_ = heads * D
    q = bf16_round(randn(B, Nq, C_, seed=70))
    k32 = randn(B, Nk, C_, seed=71)
    v = bf16_round(randn(B, Nk, C_, seed=72))
    ref = attn_ref(q, k32, v, heads)
    kpre = (k32 * (1.4426950408889634 / math.sqrt(D))).to(HDT).to(DEV)
    ldvt = (Nk + 7) // 8 * 8
    vt = torch.zeros((B, C_, ldvt), dtype=HDT, device=DEV)
    vt[:, :, :Nk] = v.permute(0, 2, 1).to(HDT).to(DEV)
    o = torch.empty(B, Nq, C_, dtype=HDT, device=DEV)
    old = L.gyre_debug_force_attn_variant(variant)    # 0 planner, 3 folded v2, 5 software-pipelined v3 (where built)
    try:
        _lib.check(L.gyre_op_attention_ex(st(), vp(to_dev_bf16(q)), C_, vp(kpre), C_, vp(vt), ldvt, B, heads, Nq, Nk, D, vp(o), C_, 1))
    finally:
        L.gyre_debug_force_attn_variant(old)
    report(f"attention prescaled v{variant} B{B} h{heads} Nq{Nq} Nk{Nk} D{D}", o.float().cpu(), ref, 6e-3)


def test_attention_prescaled_peaked_and_drifting_max():
    """Folded softmax stress: (a) one key far above the rest in a late tile (re-centre branch after tile 0),
    (b) scores that keep growing tile after tile by less than the re-centre threshold (no re-centre: large p),
    (c) all scores very negative relative to tile 0."""
    L = _lib.lib()
    B, heads, N, D = 1, 2, 640, 40
    C_ = heads * D
    c = 1.4426950408889634 / math.sqrt(D)
    q = bf16_round(randn(B, N, C_, seed=73) * 3)
    k32 = randn(B, N, C_, seed=74) * 3
    k32[:, 500] = q[:, 7] * 4                                 # (a) huge logit for query 7 in tile 7
    ramp = torch.linspace(0, 1, N).view(1, N, 1)
    k32 = k32 + q.mean(dim=1, keepdim=True) * ramp * 6        # (b) drift along the key axis
    k32[:, :64] = k32[:, :64] + 0.0
    v = bf16_round(randn(B, N, C_, seed=75))
    for name, kk in (("peaked+drift", k32), ("late keys tiny", torch.cat([k32[:, :64], k32[:, 64:] * 0.01], dim=1))):
        ref = attn_ref(q, kk, v, heads)
        kpre = (kk * c).to(HDT).to(DEV)
        vt = v.permute(0, 2, 1).to(HDT).contiguous().to(DEV)
        for variant in (3, 5):
            o = torch.empty(B, N, C_, dtype=HDT, device=DEV)
            old = L.gyre_debug_force_attn_variant(variant)
            try:
                _lib.check(L.gyre_op_attention_ex(st(), vp(to_dev_bf16(q)), C_, vp(kpre), C_, vp(vt), N, B, heads, N, N, D, vp(o), C_, 1))
            finally:
                L.gyre_debug_force_attn_variant(old)
            assert bool(torch.isfinite(o).all())
            report(f"attention prescaled v{variant} {name}", o.float().cpu(), ref, 3e-2)


@pytest.mark.parametrize("D", [32, 40, 64])
@pytest.mark.parametrize("excess", [60.0, 90.0, 300.0])
def test_attention_optimistic_pass_and_its_fallback(D, excess):
    """The pipelined kernel's first pass centres every row on the maximum of the FIRST key tile and checks no later tile.  A late
    key whose score lies `excess` (log2 units) above that: 60 - right at the acceptance bound 2^60 (= the checked pass's re-centring
    threshold TAU), 90 - finite but above the bound, 300 - exp2 overflows: the workgroup repeats the pass with the per-tile check
    (variant 7 = that pass from the start).  All must agree with the fp32 reference, and - round 6, the bound now EQUALS TAU - the
    default path must give variant 7's bits everywhere: an accepted workgroup never met a score the checked pass would re-centre on,
    a rejected one runs the checked pass.  (D = 64 keeps the checked pass: same kernel twice.)"""
    L = _lib.lib()
    B, heads, N = 1, 2, 1024
    C_ = heads * D
    c = 1.4426950408889634 / math.sqrt(D)
    q = bf16_round(randn(B, N, C_, seed=173))
    k32 = randn(B, N, C_, seed=174)
    # key 700 of head 0 aligned with query 5: its prescaled score exceeds every first-tile score of that row by about `excess`
    qn = q[0, 5, :D]
    k32[0, 700, :D] = qn * (excess / c / float(qn @ qn))
    v = bf16_round(randn(B, N, C_, seed=175))
    ref = attn_ref(q, k32, v, heads)
    kpre = (k32 * c).to(HDT).to(DEV)
    vt = v.permute(0, 2, 1).to(HDT).contiguous().to(DEV)
    outs = []
    for variant in (0, 7):                          # 0 = the default: optimistic first pass; 7 = per-tile check from the start
        o = torch.full((B, N, C_), float("nan"), dtype=HDT, device=DEV)
        old = L.gyre_debug_force_attn_variant(variant)
        try:
            _lib.check(L.gyre_op_attention_ex(st(), vp(to_dev_bf16(q)), C_, vp(kpre), C_, vp(vt), N, B, heads, N, N, D, vp(o), C_, 1))
        finally:
            L.gyre_debug_force_attn_variant(old)
        assert bool(torch.isfinite(o).all())
        report(f"attention D{D} late excess {excess} variant {variant}", o.float().cpu(), ref, 3e-2)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]), "the optimistic default must be bit-identical to the always-checked pass"


@pytest.mark.parametrize("cfg,B,tokens,C", [(0, 16, 4096, 320), (4, 4, 1024, 320), (5, 2, 1024, 320), (6, 2, 1024, 640), (7, 2, 512, 640),
                                            (0, 16, 1024, 640), (5, 3, 264, 320), (8, 2, 1024, 320), (0, 16, 256, 1280)])
def test_fused_qkv_projection(cfg, B, tokens, C):
    """Q | K | V in one GEMM launch; the V tiles go through the transposing epilogue into V^T[b][c][token]."""
    L = _lib.lib()
    M = B * tokens
    x = bf16_round(randn(M, C, seed=80))
    w = bf16_round(randn(3 * C, C, seed=81) / math.sqrt(C))
    ref = F.linear(x, w)
    qk = torch.empty(M, 2 * C, dtype=HDT, device=DEV)
    ldt = tokens
    vt = torch.full((B, C, ldt), float("nan"), dtype=HDT, device=DEV)
    old = L.gyre_debug_force_gemm_cfg(cfg)
    try:
        _lib.check(L.gyre_op_qkv(st(), vp(to_dev_bf16(x)), M, C, vp(repack_linear(w)), tokens, vp(qk), vp(vt), ldt))
    finally:
        L.gyre_debug_force_gemm_cfg(old)
    report(f"qkv cfg{cfg} QK part", qk.float().cpu(), ref[:, :2 * C], TOL)
    v_ref = ref[:, 2 * C:].reshape(B, tokens, C).permute(0, 2, 1)
    report(f"qkv cfg{cfg} V^T part", vt.float().cpu(), v_ref, TOL)


def test_fused_qkv_rejects_unaligned_configs():
    L = _lib.lib()
    x = torch.zeros(256, 320, dtype=HDT, device=DEV)
    w = torch.zeros(960, 320, dtype=HDT, device=DEV)
    qk = torch.empty(256, 640, dtype=HDT, device=DEV); vt = torch.empty(1, 320, 256, dtype=HDT, device=DEV)
    old = L.gyre_debug_force_gemm_cfg(1)     # 4-wave config: no transposing epilogue
    try:
        rc = L.gyre_op_qkv(st(), vp(x), 256, 320, vp(w), 256, vp(qk), vp(vt), 256)
    finally:
        L.gyre_debug_force_gemm_cfg(old)
    assert rc == -6
    assert L.gyre_op_qkv(st(), vp(x), 256, 320, vp(w), 100, vp(qk), vp(vt), 256) == -1   # tokens must divide M



# ---- LayerNorm folded into the consuming GEMM (GemmParams::ln_colsum; gyre_op_ln_linear) -----------------------------------
def _ln_inputs(M, K, seed, offset=0.3, scale=2.0):
    x = bf16_round(randn(M, K, seed=seed) * scale + offset)
    g, b = randn(K, seed=seed + 1) * 0.2 + 1, randn(K, seed=seed + 2) * 0.2
    return x, g, b


@pytest.mark.parametrize("M,K,N,bias", [(65536, 320, 320, True), (16384, 640, 640, False), (4096, 1280, 1280, True),
                                        (40000, 320, 640, True), (65536, 320, 960, False), (16384 + 56, 640, 1280, True)])
def test_ln_linear(M, K, N, bias):
    """y = LayerNorm(x) @ W^T + b without the normalised tensor (row statistics from one streaming pass, gamma folded into the
    weights, normalisation in the GEMM epilogue): against the fp32 reference and against the HIP path it replaces."""
    L = _lib.lib()
    x, g, b = _ln_inputs(M, K, 90)
    w = bf16_round(randn(N, K, seed=93) / math.sqrt(K))
    bias_t = randn(N, seed=94) if bias else None
    ref = F.linear(F.layer_norm(x, (K,), g, b, 1e-5), w, bias_t)
    xd, wd = to_dev_bf16(x), repack_linear(w)
    gd, bd = g.to(DEV), b.to(DEV)
    bias_d = bias_t.to(DEV) if bias else None
    ws = torch.empty(L.gyre_op_ln_linear_workspace(N, K, M), dtype=torch.uint8, device=DEV)
    y = torch.full((M, N), float("nan"), dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_ln_linear(st(), vp(xd), M, K, vp(gd), vp(bd), 1e-5, vp(wd), N, vp(bias_d) if bias else None, 0, 0, None, 0,
                                   None, 0, vp(ws), ws.numel(), vp(y)))
    report(f"ln_linear M{M} K{K} N{N}", y.float().cpu(), ref, TOL)
    n = torch.empty(M, K, dtype=HDT, device=DEV)
    y2 = torch.empty(M, N, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_layernorm(st(), vp(xd), M, K, vp(gd), vp(bd), 1e-5, vp(n)))
    _lib.check(L.gyre_op_linear(st(), vp(n), M, K, vp(wd), N, vp(bias_d) if bias else None, None, 0, vp(y2)))
    e_fused, e_two = rel_l2(y.float().cpu(), ref), rel_l2(y2.float().cpu(), ref)
    assert e_fused <= 1.25 * e_two + 1e-4, (e_fused, e_two)          # no worse than the path it replaces
    # rows are independent: the same rows inside a smaller problem give the same bits
    Ms = 4096
    if M > Ms and L.gyre_op_ln_linear(st(), vp(xd), Ms, K, vp(gd), vp(bd), 1e-5, vp(wd), N, vp(bias_d) if bias else None, 0, 0, None,
                                      0, None, 0, vp(ws), ws.numel(), vp(y2)) == 0:
        assert torch.equal(y2[:Ms], y[:Ms])


@pytest.mark.parametrize("M,K,F_", [(65536, 320, 1280), (16384, 640, 2560), (4096, 1280, 5120)])
def test_ln_linear_geglu(M, K, F_):
    L = _lib.lib()
    x, g, b = _ln_inputs(M, K, 95)
    w = bf16_round(randn(2 * F_, K, seed=98) / math.sqrt(K))
    bias_t = randn(2 * F_, seed=99) * 0.5
    val, gate = F.linear(F.layer_norm(x, (K,), g, b, 1e-5), w, bias_t).chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(2 * F_, K, M), dtype=torch.uint8, device=DEV)
    y = torch.full((M, F_), float("nan"), dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_ln_linear(st(), vp(to_dev_bf16(x)), M, K, vp(g.to(DEV)), vp(b.to(DEV)), 1e-5, vp(repack_linear(w, geglu=True)), F_,
                                   vp(repack_bias(bias_t, geglu=True)), 1, 0, None, 0, None, 0, vp(ws), ws.numel(), vp(y)))
    report(f"ln_geglu M{M} K{K} F{F_}", y.float().cpu(), ref, TOL)


@pytest.mark.parametrize("B,tokens,C", [(16, 4096, 320), (16, 1024, 640), (16, 256, 1280), (3, 1032, 320)])
def test_ln_fused_qkv(B, tokens, C):
    """norm1 -> Q | K | V of a self-attention as one launch, V leaving through the transposing epilogue."""
    L = _lib.lib()
    M = B * tokens
    x, g, b = _ln_inputs(M, C, 100)
    w = bf16_round(randn(3 * C, C, seed=103) / math.sqrt(C))
    ref = F.linear(F.layer_norm(x, (C,), g, b, 1e-5), w)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(3 * C, C, M), dtype=torch.uint8, device=DEV)
    qk = torch.full((M, 2 * C), float("nan"), dtype=HDT, device=DEV)
    vt = torch.full((B, C, tokens), float("nan"), dtype=HDT, device=DEV)
    rc = L.gyre_op_ln_linear(st(), vp(to_dev_bf16(x)), M, C, vp(g.to(DEV)), vp(b.to(DEV)), 1e-5, vp(repack_linear(w)), 3 * C, None, 0,
                             tokens, vp(vt), tokens, None, 0, vp(ws), ws.numel(), vp(qk))
    if rc == -6:
        pytest.skip("planner picks a tile config without the folded form for this shape")
    _lib.check(rc)
    report(f"ln_qkv QK part B{B} T{tokens} C{C}", qk.float().cpu(), ref[:, :2 * C], TOL)
    report(f"ln_qkv V^T part", vt.float().cpu(), ref[:, 2 * C:].reshape(B, tokens, C).permute(0, 2, 1), TOL)


def test_ln_linear_large_mean_and_small_shapes():
    """|mean| = 8 sigma (the epilogue subtracts rstd * mean * colsum from rstd * acc in fp32); shapes the planner gives to a 4-wave kernel are
    refused (GYRE_ERR_UNSUPPORTED) - the model then runs the separate LayerNorm."""
    L = _lib.lib()
    M, K, N = 65536, 320, 320
    x, g, b = _ln_inputs(M, K, 105, offset=8.0, scale=1.0)
    w = bf16_round(randn(N, K, seed=108) / math.sqrt(K))
    ref = F.linear(F.layer_norm(x, (K,), g, b, 1e-5), w)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(N, K, M), dtype=torch.uint8, device=DEV)
    y = torch.empty(M, N, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_ln_linear(st(), vp(to_dev_bf16(x)), M, K, vp(g.to(DEV)), vp(b.to(DEV)), 1e-5, vp(repack_linear(w)), N, None, 0, 0,
                                   None, 0, None, 0, vp(ws), ws.numel(), vp(y)))
    report("ln_linear mean = 8 sigma", y.float().cpu(), ref, 2 * TOL)
    xs = torch.zeros(64, 64, dtype=HDT, device=DEV); wsm = torch.zeros(64, 64, dtype=HDT, device=DEV)
    gs = torch.ones(64, device=DEV); ys = torch.empty(64, 64, dtype=HDT, device=DEV)
    assert L.gyre_op_ln_linear(st(), vp(xs), 64, 64, vp(gs), vp(gs), 1e-5, vp(wsm), 64, None, 0, 0, None, 0, None, 0, vp(ws), ws.numel(), vp(ys)) == -6
    assert L.gyre_op_ln_linear(st(), vp(xs), 64, 64, vp(gs), vp(gs), 1e-5, vp(wsm), 64, None, 0, 0, None, 0, None, 0, vp(ws), 16, vp(ys)) == -4


@pytest.mark.parametrize("M,C,res", [(65536, 320, True), (16384, 640, True), (4096, 1280, True), (40000 + 24, 320, False),
                                     (16384, 1280, True)])
def test_linear_row_statistics_feed_the_folded_layernorm(M, C, res):
    """A C x C projection (+ residual) leaves per row the partial sums of its rounded outputs (one per N tile); the next GEMM's
    folded LayerNorm finishes mean / rstd from them - no statistics pass.  Checks the partial sums against the stored tensor,
    and the chained result against the fp32 reference and against the same chain with the statistics pass (same bits unless
    E[x^2] - mean^2 and the two-pass variance round differently: compared by tolerance)."""
    L = _lib.lib()
    parts = L.gyre_op_linear_rowstats_parts(M, C, C, 1 if res else 0)
    if parts <= 0:
        pytest.skip("planner picks a tile config without the row-statistics epilogue for this shape")
    x = bf16_round(randn(M, C, seed=110))
    w1 = bf16_round(randn(C, C, seed=111) / math.sqrt(C))
    b1 = randn(C, seed=112) * 0.3 + 0.2
    r = bf16_round(randn(M, C, seed=113) + 0.5) if res else None
    y1 = torch.empty(M, C, dtype=HDT, device=DEV)
    stats = torch.full((parts, M, 2), float("nan"), device=DEV)
    _lib.check(L.gyre_op_linear_rowstats(st(), vp(to_dev_bf16(x)), M, C, vp(repack_linear(w1)), C, vp(b1.to(DEV)),
                                         vp(to_dev_bf16(r)) if res else None, vp(y1), vp(stats)))
    ref1 = F.linear(x, w1, b1) + (r if res else 0)
    report(f"linear+rowstats M{M} C{C}", y1.float().cpu(), ref1, TOL)
    y1f = y1.float()
    tot = stats.sum(0)
    assert torch.allclose(tot[:, 0], y1f.sum(1), rtol=1e-4, atol=2e-3)
    assert torch.allclose(tot[:, 1], (y1f * y1f).sum(1), rtol=1e-4, atol=2e-3)
    # consumer: LayerNorm(y1) @ w2^T with gamma / beta, statistics from the partial sums
    g, b = randn(C, seed=114) * 0.2 + 1, randn(C, seed=115) * 0.2
    w2 = bf16_round(randn(C, C, seed=116) / math.sqrt(C))
    ref2 = F.linear(F.layer_norm(y1f.cpu(), (C,), g, b, 1e-5), w2)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(C, C, M), dtype=torch.uint8, device=DEV)
    out_p = torch.empty(M, C, dtype=HDT, device=DEV)
    out_s = torch.empty(M, C, dtype=HDT, device=DEV)
    args = (st(), vp(y1), M, C, vp(g.to(DEV)), vp(b.to(DEV)), 1e-5, vp(repack_linear(w2)), C, None, 0, 0, None, 0)
    _lib.check(L.gyre_op_ln_linear(*args, vp(stats), parts, vp(ws), ws.numel(), vp(out_p)))
    _lib.check(L.gyre_op_ln_linear(*args, None, 0, vp(ws), ws.numel(), vp(out_s)))
    report(f"ln_linear from producer statistics M{M} C{C}", out_p.float().cpu(), ref2, TOL)
    assert rel_l2(out_p.float().cpu(), out_s.float().cpu()) < 2e-3
    if M == 4096:
        # weight-dominated GEGLU consumer (N = 8C > M): the planner gives it the pipelined 256x320 kernel, whose epilogue
        # finishes the statistics from the same partial sums
        F_ = 4 * C
        w3 = bf16_round(randn(2 * F_, C, seed=117) / math.sqrt(C))
        b3 = randn(2 * F_, seed=118) * 0.5
        val, gate = F.linear(F.layer_norm(y1f.cpu(), (C,), g, b, 1e-5), w3, b3).chunk(2, dim=-1)
        ws3 = torch.empty(L.gyre_op_ln_linear_workspace(2 * F_, C, M), dtype=torch.uint8, device=DEV)
        gp = torch.empty(M, F_, dtype=HDT, device=DEV)
        gs = torch.empty(M, F_, dtype=HDT, device=DEV)
        gargs = (st(), vp(y1), M, C, vp(g.to(DEV)), vp(b.to(DEV)), 1e-5, vp(repack_linear(w3, geglu=True)), F_,
                 vp(repack_bias(b3, geglu=True)), 1, 0, None, 0)
        _lib.check(L.gyre_op_ln_linear(*gargs, vp(stats), parts, vp(ws3), ws3.numel(), vp(gp)))
        _lib.check(L.gyre_op_ln_linear(*gargs, None, 0, vp(ws3), ws3.numel(), vp(gs)))
        report(f"ln_geglu from producer statistics M{M} C{C}", gp.float().cpu(), val * F.gelu(gate), TOL)
        assert rel_l2(gp.float().cpu(), gs.float().cpu()) < 2e-3


# ---- pipelined 32x32x16 tile configs (kernels_gemm4s.hip): 4 waves 20 = 192x320, 21 = 256x256, 22 = 128x320, 23 = 128x256;
# 8 waves 24 = 256x320 ----------
def _ablation(bits):
    return _lib.lib().gyre_debug_gemm_ablation(bits)


@pytest.mark.parametrize("cfg", [20, 21, 22, 23, 24])
@pytest.mark.parametrize("M,K,N,bias,res", [(1000, 320, 640, True, True), (4096 + 37, 1280, 1280, True, False),
                                            (513, 2560, 960, False, True), (192, 64, 320, True, False)])
def test_linear_4s_tile_config(cfg, M, K, N, bias, res):
    L = _lib.lib()
    x = bf16_round(randn(M, K, seed=70))
    w = bf16_round(randn(N, K, seed=71) / math.sqrt(K))
    b = randn(N, seed=72) if bias else None
    r = bf16_round(randn(M, N, seed=73)) if res else None
    ref = F.linear(x, w, b)
    if res:
        ref = ref + r
    y = torch.empty(M, N, dtype=HDT, device=DEV)
    old = L.gyre_debug_force_gemm_cfg(cfg)
    try:
        _lib.check(L.gyre_op_linear(st(), vp(to_dev_bf16(x)), M, K, vp(repack_linear(w)), N,
                                    vp(b.to(DEV)) if bias else None, vp(to_dev_bf16(r)) if res else None, 0, vp(y)))
    finally:
        L.gyre_debug_force_gemm_cfg(old)
    report(f"linear 4s cfg{cfg} M{M} K{K} N{N}", y.float().cpu(), ref, TOL)


@pytest.mark.parametrize("cfg", [20, 21, 22, 23, 24])
def test_geglu_4s_tile_config(cfg):
    L = _lib.lib()
    M, K, F_ = 700, 320, 1280
    x = bf16_round(randn(M, K, seed=44))
    w = bf16_round(randn(2 * F_, K, seed=45) / math.sqrt(K))
    b = randn(2 * F_, seed=46) * 0.5
    val, gate = F.linear(x, w, b).chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    y = torch.empty(M, F_, dtype=HDT, device=DEV)
    old = L.gyre_debug_force_gemm_cfg(cfg)
    try:
        _lib.check(L.gyre_op_linear(st(), vp(to_dev_bf16(x)), M, K, vp(repack_linear(w, geglu=True)), F_,
                                    vp(repack_bias(b, geglu=True)), None, 1, vp(y)))
    finally:
        L.gyre_debug_force_gemm_cfg(old)
    report(f"geglu 4s cfg{cfg}", y.float().cpu(), ref, TOL)


@pytest.mark.parametrize("zfill", [0, 1])
@pytest.mark.parametrize("cfg", [20, 21, 22, 23, 24])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups,asym,res", [
    (2, 24, 20, 320, 640, 1, 0, 0, True), (1, 16, 16, 128, 1280, 1, 1, 0, False), (2, 18, 18, 128, 1280, 2, 0, 1, False),
    (3, 8, 8, 64, 320, 1, 0, 0, False), (1, 19, 21, 192, 256, 1, 0, 0, True),
])
def test_conv_4s_tile_config(zfill, cfg, B, H, W, Cin, Cout, stride, ups, asym, res):
    """zfill 0: conv zero padding through out-of-range raw-buffer LDS-DMA requests; 1: through zero-page pointers."""
    L = _lib.lib()
    x = bf16_round(randn(B, Cin, H, W, seed=47))
    w = bf16_round(randn(Cout, Cin, 3, 3, seed=48) / math.sqrt(9 * Cin))
    b = randn(Cout, seed=49)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(F.pad(xi, (0, 1, 0, 1)), w, b, stride=stride, padding=0) if asym else F.conv2d(xi, w, b, stride=stride, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    r = bf16_round(randn(B, Cout, Ho, Wo, seed=50)) if res else None
    if res:
        ref = ref + r
    y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=HDT, device=DEV)
    old = L.gyre_debug_force_gemm_cfg(cfg)
    olda = _ablation(0x200 if zfill else 0)
    try:
        _lib.check(L.gyre_op_conv3x3(st(), vp(to_dev_bf16(nhwc(x))), B, H, W, Cin, vp(repack_conv(w)), Cout, vp(b.to(DEV)),
                                     vp(to_dev_bf16(nhwc(r))) if res else None, stride, ups, asym, vp(y)))
    finally:
        L.gyre_debug_force_gemm_cfg(old)
        _ablation(olda)
    report(f"conv 4s cfg{cfg} zfill{zfill} {B}x{H}x{W} {Cin}->{Cout}", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)


@pytest.mark.parametrize("cfg,splits", [(20, 3), (21, 4), (22, 8), (23, 5), (24, 6)])
def test_conv_4s_split_k(cfg, splits):
    L = _lib.lib()
    B, H, W, Cin, Cout = 4, 16, 16, 640, 1280
    x = bf16_round(randn(B, Cin, H, W, seed=51))
    w = bf16_round(randn(Cout, Cin, 3, 3, seed=52) / math.sqrt(9 * Cin))
    b = randn(Cout, seed=53)
    r = bf16_round(randn(B, Cout, H, W, seed=54))
    ref = F.conv2d(x, w, b, padding=1) + r
    y = torch.empty(B, H, W, Cout, dtype=HDT, device=DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    L.gyre_debug_set_splitk_workspace(vp(ws), ws.numel())
    old = L.gyre_debug_force_gemm_cfg(cfg | (splits << 8))
    try:
        args = (st(), vp(to_dev_bf16(nhwc(x))), B, H, W, Cin, vp(repack_conv(w)), Cout, vp(b.to(DEV)), vp(to_dev_bf16(nhwc(r))), 1, 0, 0, vp(y))
        _lib.check(L.gyre_op_conv3x3(*args))
        y1 = y.clone()
        _lib.check(L.gyre_op_conv3x3(*args))
        assert torch.equal(y, y1)
    finally:
        L.gyre_debug_force_gemm_cfg(old)
        L.gyre_debug_set_splitk_workspace(None, 0)
    report(f"conv 4s split-K cfg{cfg} x{splits}", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)


def test_4s_tile_configs_sum_in_the_same_order_as_the_others():
    """32x32x16 fragments, four k16 sub-steps per 64-channel chunk: same bits as the 16x16x32 kernels."""
    L = _lib.lib()
    B, H, W, Cin, Cout = 2, 20, 24, 192, 1280
    x = to_dev_bf16(nhwc(bf16_round(randn(B, Cin, H, W, seed=60))))
    w = repack_conv(bf16_round(randn(Cout, Cin, 3, 3, seed=61) / math.sqrt(9 * Cin)))
    b = randn(Cout, seed=62).to(DEV)
    outs = {}
    for cfg in (1, 4, 20, 21, 22, 23, 24):
        y = torch.empty(B, H, W, Cout, dtype=HDT, device=DEV)
        old = L.gyre_debug_force_gemm_cfg(cfg)
        try:
            _lib.check(L.gyre_op_conv3x3(st(), vp(x), B, H, W, Cin, vp(w), Cout, vp(b), None, 1, 0, 0, vp(y)))
        finally:
            L.gyre_debug_force_gemm_cfg(old)
        outs[cfg] = y
    for cfg, y in outs.items():
        assert torch.equal(y, outs[1]), f"conv: config {cfg} differs bitwise from config 1"
    M, K, N = 777, 640, 1280
    xl = to_dev_bf16(bf16_round(randn(M, K, seed=63)))
    wl = repack_linear(bf16_round(randn(N, K, seed=64) / math.sqrt(K)))
    outs = {}
    for cfg in (1, 4, 20, 21, 22, 23, 24):
        y = torch.empty(M, N, dtype=HDT, device=DEV)
        old = L.gyre_debug_force_gemm_cfg(cfg)
        try:
            _lib.check(L.gyre_op_linear(st(), vp(xl), M, K, vp(wl), N, None, None, 0, vp(y)))
        finally:
            L.gyre_debug_force_gemm_cfg(old)
        outs[cfg] = y
    for cfg, y in outs.items():
        assert torch.equal(y, outs[1]), f"linear: config {cfg} differs bitwise from config 1"


def test_4s_rejects_unsupported_shapes():
    L = _lib.lib()
    x = to_dev_bf16(bf16_round(randn(256, 200, seed=1)))
    w = repack_linear(bf16_round(randn(320, 200, seed=2)))
    y = torch.empty(256, 320, dtype=HDT, device=DEV)
    old = L.gyre_debug_force_gemm_cfg(20)
    try:
        assert L.gyre_op_linear(st(), vp(x), 256, 200, vp(w), 320, None, None, 0, vp(y)) == -6   # K not in 64-channel steps
    finally:
        L.gyre_debug_force_gemm_cfg(old)


# ---- ToMe: bipartite soft matching + K / V merge (kernels_tome.hip) vs oracle/tome_ref.py -----------------------------------
def _tome_run(k, v, r):
    L = _lib.lib()
    B, N, Cc = k.shape
    reff = min(r, N // 2)
    ldvt = (N - reff + 7) // 8 * 8
    kd, vd = to_dev_bf16(k), to_dev_bf16(v)
    ws = torch.empty(L.gyre_op_tome_workspace(B, N, Cc), dtype=torch.uint8, device=DEV)
    k_out = torch.empty(B, N - reff, Cc, dtype=HDT, device=DEV)
    vt_out = torch.full((B, Cc, ldvt), float("nan"), dtype=HDT, device=DEV)
    order = torch.empty(B, N // 2, dtype=torch.int32, device=DEV)
    nidx = torch.empty(B, N // 2, dtype=torch.int32, device=DEV)
    _lib.check(L.gyre_op_tome_merge(st(), vp(kd), Cc, vp(vd), Cc, B, N, Cc, r, vp(ws), ws.numel(), vp(k_out), vp(vt_out), ldvt,
                                    vp(order), vp(nidx)))
    torch.cuda.synchronize()
    return k_out.float().cpu(), vt_out.float().cpu(), order.cpu().long(), nidx.cpu().long(), reff


@pytest.mark.parametrize("B,N,C,r", [(2, 256, 320, 64), (3, 1024, 640, 512), (1, 4096, 320, 1000), (2, 64, 1280, 40), (2, 136, 64, 17)])
def test_tome_merge_matches_oracle(B, N, C, r):
    """(a) the merge arithmetic: with the kernel's OWN selection (order / node_idx it returns) the oracle's merge_wavg gives the
    same K' and V'^T up to one bf16 rounding; (b) the selection: on data with unambiguous partners the kernel picks exactly
    the oracle's matches and ranking; on Gaussian data (many near-ties after the bf16 rounding of the scores) at least 97 %."""
    from oracle import tome_ref as TR
    g = torch.Generator().manual_seed(100 + N)
    k = bf16_round(torch.randn(B, N, C, generator=g))
    v = bf16_round(torch.randn(B, N, C, generator=g))
    ko, vto, order, nidx, reff = _tome_run(k, v, r)
    assert ko.shape == (B, N - reff, C)
    kref, vref = TR.merge_wavg(k, order, nidx, reff), TR.merge_wavg(v, order, nidx, reff)
    report(f"tome merged K B{B} N{N} C{C} r{r}", ko, kref, TOL)
    report(f"tome merged V^T B{B} N{N} C{C} r{r}", vto[:, :, :N - reff], vref.transpose(1, 2), TOL)
    assert float(vto[:, :, N - reff:].abs().max() if vto.shape[2] > N - reff else 0.0) == 0.0       # padding columns are zero
    for b in range(B):                                        # order is a permutation of the a tokens, matches are b tokens
        assert sorted(order[b].tolist()) == list(range(N // 2)) and int(nidx[b].min()) >= 0 and int(nidx[b].max()) < N // 2
    o_ref, n_ref, _ = TR.bipartite_soft_matching(k, r)
    agree = float((nidx == n_ref).float().mean())
    print(f"[parity] tome random data: best-match agreement {agree:.4f}")
    assert agree >= 0.97
    # unambiguous partners: a token i is a noisy copy of b token perm[i], noise level grows with i -> strict ranking
    half = N // 2
    perm = torch.stack([torch.randperm(half, generator=g) for _ in range(B)])
    bt = torch.randn(B, half, C, generator=g)
    noise = torch.randn(B, half, C, generator=g) * (0.02 + 0.9 * torch.arange(half)[None, :, None] / half)
    at = torch.gather(bt, 1, perm[:, :, None].expand(-1, -1, C)) + noise
    ks = torch.zeros(B, 2 * half, C)
    ks[:, 0::2], ks[:, 1::2] = at, bt
    ks = bf16_round(torch.cat([ks, torch.randn(B, N - 2 * half, C, generator=g)], 1))
    _, _, order2, nidx2, _ = _tome_run(ks, ks, r)
    o_ref, n_ref, _ = TR.bipartite_soft_matching(ks, r)
    assert float((nidx2[:, : half // 2] == perm[:, : half // 2]).float().mean()) == 1.0           # clean copies find their source
    assert float((nidx2 == n_ref).float().mean()) >= 0.995
    top = min(reff, half // 4)
    for b in range(B):
        assert len(set(order2[b, :top].tolist()) & set(o_ref[b, :top].tolist())) >= 0.95 * top


def test_tome_rejects_bad_arguments():
    L = _lib.lib()
    x = torch.zeros(1, 64, 36, dtype=HDT, device=DEV)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    out = torch.empty(1, 64, 36, dtype=HDT, device=DEV)
    assert L.gyre_op_tome_merge(st(), vp(x), 36, vp(x), 36, 1, 64, 36, 8, vp(ws), ws.numel(), vp(out), vp(out), 64, None, None) == -1   # C % 8
    x = torch.zeros(1, 64, 64, dtype=HDT, device=DEV)
    assert L.gyre_op_tome_merge(st(), vp(x), 64, vp(x), 64, 1, 64, 64, 8, vp(ws), 16, vp(out), vp(out), 64, None, None) == -4         # workspace


# ---- GroupNorm statistics from the producing kernel (GemmParams::colstat_out) -------------------------------------------------
def _colstats_ref(y_nhwc, B, rows, unit):
    """[B * HW / rows][C / unit][2] sums / sums of squares of the stored bf16 values, in float64."""
    C = y_nhwc.shape[-1]
    t = y_nhwc.double().reshape(-1, rows, C // unit, unit)
    return torch.stack([t.sum(dim=(1, 3)), (t * t).sum(dim=(1, 3))], dim=-1)


def _check_colstats(name, stats, y, B, rows, unit):
    ref = _colstats_ref(y.float().cpu().reshape(-1, y.shape[-1]), B, rows, unit)
    got = stats.double().cpu().reshape(ref.shape)
    scale = ref[..., 1].sqrt().mean() * math.sqrt(rows * unit)          # typical magnitude of a sum of rows*unit values
    e_sum = float((got[..., 0] - ref[..., 0]).abs().max() / scale)
    e_sq = float(((got[..., 1] - ref[..., 1]).abs() / ref[..., 1].clamp_min(1e-30)).max())
    print(f"[parity] {name}: column statistics vs float64 sums of the stored values: sum err {e_sum:.2e} (of a typical sum), "
          f"sumsq rel err {e_sq:.2e}, row block {rows}")
    assert e_sum < 1e-4 and e_sq < 1e-4


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups,res,want_rows", [
    (16, 64, 64, 320, 320, 1, 0, True, 256),      # pipelined 256x320 tile, unsplit (64x64 level resnet convs)
    (2, 64, 64, 8, 320, 1, 0, False, 0),          # conv_in-like (K = 72): whatever the planner picks, or unsupported
    (16, 32, 32, 640, 640, 1, 0, True, 16),       # split K: statistics from the reduction kernel
    (16, 64, 64, 320, 320, 2, 0, False, 0),       # downsample conv -> 32x32
    (16, 32, 32, 640, 640, 1, 1, False, 256),     # upsample conv -> 64x64
    (16, 32, 32, 1920, 640, 1, 0, True, 0),
    (8, 64, 64, 320, 320, 1, 0, True, 128),       # 128x160 tile, unsplit (the shared prefix of a CFG batch runs at half the batch)
    (8, 64, 64, 320, 320, 2, 0, False, 0),        # downsample conv at half the batch
    (16, 32, 32, 320, 640, 1, 0, False, 0),
    (2, 64, 64, 320, 320, 1, 0, True, 64),        # split K at the 64x64 level: 64-row blocks (at most 64 per sample)
])
def test_conv_emits_groupnorm_statistics(B, H, W, Cin, Cout, stride, ups, res, want_rows):
    import ctypes as C
    L = _lib.lib()
    unit = 10
    x = bf16_round(randn(B, Cin, H, W, seed=61) + 0.3)
    w = bf16_round(randn(Cout, Cin, 3, 3, seed=62) / math.sqrt(9 * Cin))
    b = randn(Cout, seed=63)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xi, w, b, stride=stride, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    r = bf16_round(randn(B, Cout, Ho, Wo, seed=64)) if res else None
    if res:
        ref = ref + r
    y = torch.empty(B, Ho, Wo, Cout, dtype=HDT, device=DEV)
    stats = torch.full((B * Ho * Wo // 16, Cout // unit, 2), float("nan"), dtype=torch.float32, device=DEV)
    need = L.gyre_op_gemm_splitk_bytes(1, B * Ho * Wo, Cout, 9 * Cin, B)
    ws = torch.empty(max(need, 16) + 256, dtype=torch.uint8, device=DEV)
    rows = C.c_int(0)
    args = (st(), vp(to_dev_bf16(nhwc(x))), B, H, W, Cin, vp(repack_conv(w)), Cout, vp(b.to(DEV)),
            vp(to_dev_bf16(nhwc(r))) if res else None, stride, ups, unit, vp(y), vp(stats), stats.numel() * 4, vp(ws), ws.numel(),
            C.byref(rows))
    rc = L.gyre_op_conv3x3_colstats(*args)
    if rc == -6 and want_rows == 0:
        assert rows.value <= 0
        pytest.skip("planner's kernel for this shape emits no column statistics: the consumer keeps its own pass")
    _lib.check(rc)
    if want_rows:
        assert rows.value == want_rows
    name = f"conv {B}x{H}x{W} {Cin}->{Cout} s{stride} ups{ups}"
    report(name, y.float().cpu().permute(0, 3, 1, 2), ref, TOL)
    nblk = B * Ho * Wo // rows.value
    _check_colstats(name, stats[:nblk], y, B, rows.value, unit)
    # the same launch without statistics stores the same bits; repeated launches are deterministic
    y2 = torch.empty_like(y)
    L.gyre_debug_set_splitk_workspace(vp(ws), ws.numel())
    try:
        _lib.check(L.gyre_op_conv3x3(st(), vp(to_dev_bf16(nhwc(x))), B, H, W, Cin, vp(repack_conv(w)), Cout, vp(b.to(DEV)),
                                     vp(to_dev_bf16(nhwc(r))) if res else None, stride, ups, 0, vp(y2)))
    finally:
        L.gyre_debug_set_splitk_workspace(None, 0)
    assert torch.equal(y, y2)
    s1 = stats[:nblk].clone()
    _lib.check(L.gyre_op_conv3x3_colstats(*args))
    assert torch.equal(stats[:nblk], s1)


@pytest.mark.parametrize("B,HW,C,res", [(16, 4096, 320, True), (16, 1024, 640, True), (2, 4096, 320, True), (3, 1024, 1280, False),
                                          (8, 4096, 320, True), (8, 1024, 640, True)])
def test_linear_emits_groupnorm_statistics(B, HW, C, res):
    """The transformer's proj_out (+ residual) feeds the next resnet's GroupNorm."""
    import ctypes as Ct
    L = _lib.lib()
    unit, M = 10, B * HW
    x = bf16_round(randn(M, C, seed=71))
    w = bf16_round(randn(C, C, seed=72) / math.sqrt(C))
    b = randn(C, seed=73)
    r = bf16_round(randn(M, C, seed=74) * 2 + 0.5) if res else None
    ref = F.linear(x, w, b) + (r if res else 0)
    y = torch.empty(M, C, dtype=HDT, device=DEV)
    stats = torch.full((M // 16, C // unit, 2), float("nan"), dtype=torch.float32, device=DEV)
    rows = Ct.c_int(0)
    rc = L.gyre_op_linear_colstats(st(), vp(to_dev_bf16(x)), M, C, vp(to_dev_bf16(w)), C, vp(b.to(DEV)),
                                   vp(to_dev_bf16(r)) if res else None, HW, unit, vp(y), vp(stats), stats.numel() * 4, None, 0,
                                   Ct.byref(rows))
    if rc == -6:
        pytest.skip("planner's kernel for this shape emits no column statistics")
    _lib.check(rc)
    assert rows.value in (16, 64, 128, 256) and HW % rows.value == 0
    report(f"linear {M}x{C}x{C}", y.float().cpu(), ref, TOL)
    _check_colstats(f"linear {M}x{C}x{C}", stats[:M // rows.value], y, B, rows.value, unit)


@pytest.mark.parametrize("B,H,W,C1,C2,silu", [(2, 64, 64, 320, 0, 1), (16, 32, 32, 640, 0, 0), (2, 32, 32, 1280, 640, 1),
                                              (2, 64, 64, 640, 320, 1), (2, 64, 64, 320, 320, 1)])
def test_groupnorm_from_producer_statistics(B, H, W, C1, C2, silu):
    """Consumer side: mean / rstd finished from per-(row block, 10-channel unit) partials of x (and of the skip tensor x2, whose
    row blocks may differ: 256-row tiles vs the 16-row blocks of a split-K reduction) equal the kernel's own statistics pass."""
    L = _lib.lib()
    unit, HW, C = 10, H * W, C1 + C2
    x = bf16_round(randn(B, C, H, W, seed=81) * 1.5 + 0.7)
    gamma, beta = randn(C, seed=82) * 0.2 + 1.0, randn(C, seed=83) * 0.3
    ref = F.group_norm(x, 32, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    xn = nhwc(x)
    a = to_dev_bf16(xn[..., :C1])
    b = to_dev_bf16(xn[..., C1:]) if C2 else None
    rows_a, rows_b = 256, 16
    cs_a = _colstats_ref(xn[..., :C1].reshape(-1, C1), B, rows_a, unit).float().to(DEV).contiguous()
    cs_b = _colstats_ref(xn[..., C1:].reshape(-1, C2), B, rows_b, unit).float().to(DEV).contiguous() if C2 else None
    ws = torch.empty(L.gyre_op_groupnorm_workspace(B, HW, C, 32) + 256, dtype=torch.uint8, device=DEV)
    y = torch.empty(B, H, W, C, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_groupnorm_colstats(st(), vp(a), vp(b), C1, B, HW, C, 32, vp(gamma.to(DEV)), vp(beta.to(DEV)), 1e-5, silu,
                                            vp(cs_a), HW // rows_a, vp(cs_b), HW // rows_b if C2 else 0, unit, vp(ws), ws.numel(), vp(y)))
    report(f"groupnorm from producer statistics {B}x{H}x{W} {C1}+{C2}", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)
    y0 = torch.empty_like(y)
    _lib.check(L.gyre_op_groupnorm(st(), vp(a), vp(b), C1, B, HW, C, 32, vp(gamma.to(DEV)), vp(beta.to(DEV)), 1e-5, silu, vp(ws),
                                   ws.numel(), vp(y0)))
    d = rel_l2(y.float(), y0.float())
    print(f"[property] vs the kernel's own statistics pass: rel-L2 {d:.2e}")
    assert d < 1e-3


def test_unet_with_and_without_producer_statistics_full_size():
    """SD1.5 UNet, 64x64 latents: GroupNorm statistics from the producing conv / GEMM epilogues (default) against the separate
    statistics pass (tuning bit 17) - same network, two summation orders of the same sums; fewer launches."""
    from gyre_amd import config as gcfg, weights
    from gyre_amd.modules import GyreHipUNet
    L = _lib.lib()
    cfg = gcfg.sd15_unet()
    net = GyreHipUNet(cfg)
    net.load_state_dict(weights.synthetic_state_dict(weights.unet_param_shapes(cfg)))
    net = net.to(HDT).to(DEV)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 64, 64, generator=g).to(DEV)
    ctx = torch.randn(2, 77, 768, generator=g).to(DEV)
    t = torch.tensor([981, 20], device=DEV)
    a = net(x, t, encoder_hidden_states=ctx).sample          # (first call: also projects the text context)
    a2 = net(x, t, encoder_hidden_states=ctx).sample
    n_a = L.gyre_last_launch_count()
    assert torch.equal(a, a2)
    old = L.gyre_debug_gemm_ablation(0x20000)
    try:
        b = net(x, t, encoder_hidden_states=ctx).sample
        n_b = L.gyre_last_launch_count()
    finally:
        L.gyre_debug_gemm_ablation(old)
    d = rel_l2(a.float(), b.float())
    print(f"[property] UNet eps with producer statistics vs separate passes: rel-L2 {d:.2e}; launches {n_a} vs {n_b}")
    assert d < 1.5e-2 and n_a < n_b
    # batch equivariance survives: permuting the samples permutes the result bit-exactly
    p = net(x.flip(0).contiguous(), t.flip(0).contiguous(), encoder_hidden_states=ctx.flip(0).contiguous()).sample
    assert torch.equal(p.flip(0), a)


def test_colstats_entry_points_reject_bad_integers():
    """rows_per_sample = 0 used to reach an integer division (SIGFPE); now every size is checked up front."""
    import ctypes as C
    L = _lib.lib()
    x = torch.zeros(256, 320, dtype=HDT, device=DEV)
    w = torch.zeros(320, 320, dtype=HDT, device=DEV)
    y = torch.zeros(256, 320, dtype=HDT, device=DEV)
    stats = torch.zeros(4096, dtype=torch.float32, device=DEV)
    rows = C.c_int(0)
    for rps, unit in ((0, 10), (256, 0), (100, 10), (256, 7)):
        rc = L.gyre_op_linear_colstats(st(), vp(x), 256, 320, vp(w), 320, None, None, rps, unit, vp(y), vp(stats), stats.numel() * 4,
                                       None, 0, C.byref(rows))
        assert rc == -1, (rps, unit, rc)
    assert L.gyre_op_gemm_splitk_bytes(1, 0, 320, 2880, 1) == 0 and L.gyre_op_gemm_splitk_bytes(1, 4096, 320, 2881, 1) == 0


# ---- the cross-attention block as one launch (kernels_xattn.hip; gyre_op_cross_attention_block) -----------------------------------
@pytest.mark.parametrize("B,Nk,bias", [(8, 77, True), (8, 80, False), (16, 33, True), (8, 1, True)])
def test_fused_cross_attention_block_operator(B, Nk, bias):
    """out = softmax(LayerNorm(x) Wq^T K^T / sqrt(d)) V Wo^T + bo + x at SD1.x's 64x64 shape (C = 320, 8 heads of 40, 4096 tokens per
    sample) against fp32 ATen on storage-rounded operands, for a full text chunk, the largest key count the kernel takes, a key count
    that leaves whole key fragments empty, and a single key; plus the per-row (sum, sum of squares) of the rounded outputs that the
    next folded LayerNorm consumes, and the refusals outside the kernel's domain."""
    L = _lib.lib()
    C_, heads, tokens = 320, 8, 4096
    D = C_ // heads
    M = B * tokens
    x = bf16_round(randn(M, C_, seed=301) * 1.5 + 0.2)
    g, be = randn(C_, seed=302) * 0.2 + 1, randn(C_, seed=303) * 0.2
    wq = bf16_round(randn(C_, C_, seed=304) / math.sqrt(C_))
    wo = bf16_round(randn(C_, C_, seed=305) / math.sqrt(C_))
    bo = randn(C_, seed=306) * 0.1 if bias else None
    k = bf16_round(randn(B, Nk, C_, seed=307))
    v = bf16_round(randn(B, Nk, C_, seed=308))
    # reference (fp32): the module chain of diffusers' BasicTransformerBlock.attn2 on norm2(x)
    xn = F.layer_norm(x, (C_,), g, be, 1e-5)
    q = F.linear(xn, wq).reshape(B, tokens, C_)
    ref = F.linear(attn_ref(q, k, v, heads).reshape(M, C_), wo, bo) + x
    # device operands: K prescaled by log2(e) / sqrt(D) (what the weight repack folds into to_k), V transposed and padded to 8 keys
    ldvt = (Nk + 7) // 8 * 8
    kpre = (k * (1.4426950408889634 / math.sqrt(D))).to(HDT).contiguous().to(DEV)
    vt = torch.full((B, C_, ldvt), float("nan"), dtype=HDT, device=DEV)              # padding = NaN on purpose
    vt[:, :, :Nk] = v.permute(0, 2, 1).to(HDT).to(DEV)
    out = torch.empty(M, C_, dtype=HDT, device=DEV)
    stats = torch.empty(M, 2, dtype=torch.float32, device=DEV)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(C_, C_, M), dtype=torch.uint8, device=DEV)
    _lib.check(L.gyre_op_cross_attention_block(st(), vp(to_dev_bf16(x)), M, tokens, C_, heads, vp(g.to(DEV)), vp(be.to(DEV)), 1e-5,
                                               vp(repack_linear(wq)), vp(kpre), vp(vt), Nk, ldvt, vp(repack_linear(wo)),
                                               vp(bo.to(DEV)) if bias else None, vp(ws), ws.numel(), vp(out), vp(stats)))
    got = out.float().cpu()
    assert bool(torch.isfinite(got).all())
    report(f"fused cross-attention block B{B} Nk{Nk}", got, ref, 8e-3)
    s_ref = torch.stack([got.sum(1), (got * got).sum(1)], dim=1)                    # statistics of what was stored
    assert torch.allclose(stats.cpu(), s_ref, rtol=2e-4, atol=2e-2)
    # outside the domain: too few rows for a full grid, too many keys, rows that straddle samples -> GYRE_ERR_UNSUPPORTED (-6)
    for (M_, tok_, nk_) in ((4 * tokens, tokens, Nk), (M, tokens, 81), (M, 4000, Nk)):
        rc = L.gyre_op_cross_attention_block(st(), vp(to_dev_bf16(x)), M_, tok_, C_, heads, vp(g.to(DEV)), vp(be.to(DEV)), 1e-5,
                                             vp(repack_linear(wq)), vp(kpre), vp(vt), nk_, ldvt, vp(repack_linear(wo)), None,
                                             vp(ws), ws.numel(), vp(out), None)
        assert rc in (-6, -1), rc


# ---- a resnet's conv2 with its 1x1 shortcut folded in (GemmParams::sc_*; gyre_op_conv3x3_shortcut) -----------------------------------
@pytest.mark.parametrize("B,H,Cin,Cout,C1,C2", [(16, 64, 320, 320, 320, 320), (16, 64, 320, 320, 640, 320), (16, 32, 640, 640, 320, 0),
                                                (16, 32, 640, 640, 1280, 640), (16, 16, 1280, 1280, 1280, 1280)])
def test_conv3x3_with_folded_shortcut(B, H, Cin, Cout, C1, C2):
    """y = conv3x3(h) + conv1x1(cat[x, skip]) in one launch: the shortcut's channels are extra K steps of the pipelined convolution
    (one fp32 accumulation, one rounding) - against fp32 ATen on storage-rounded operands, for the shapes of SD1.5's up path (two
    shortcut sources: the concatenation is never materialised), a down-path resnet (one source) and a deep level that runs in K slices."""
    L = _lib.lib()
    h = bf16_round(randn(B, Cin, H, H, seed=401))
    x1 = bf16_round(randn(B, C1, H, H, seed=402))
    x2 = bf16_round(randn(B, C2, H, H, seed=403)) if C2 else None
    w = bf16_round(randn(Cout, Cin, 3, 3, seed=404) / math.sqrt(9 * Cin))
    wsc = bf16_round(randn(Cout, C1 + C2, seed=405) / math.sqrt(C1 + C2))
    b, bsc = randn(Cout, seed=406) * 0.1, randn(Cout, seed=407) * 0.1
    xcat = x1 if x2 is None else torch.cat([x1, x2], dim=1)
    ref = F.conv2d(h, w, b, padding=1) + F.conv2d(xcat, wsc[:, :, None, None], bsc)
    y = torch.empty(B, H, H, Cout, dtype=HDT, device=DEV)
    ws = torch.empty(Cout * (9 * Cin + C1 + C2) * 2 + Cout * 4 + 512, dtype=torch.uint8, device=DEV)
    wsk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)                  # split-K slabs of the deep level
    L.gyre_debug_set_splitk_workspace(vp(wsk), wsk.numel())
    try:
        rc = L.gyre_op_conv3x3_shortcut(st(), vp(to_dev_bf16(nhwc(h))), B, H, H, Cin, vp(repack_conv(w)), Cout, vp(repack_bias(b)),
                                        vp(to_dev_bf16(nhwc(x1))), C1, vp(to_dev_bf16(nhwc(x2))) if C2 else None, C2,
                                        vp(repack_linear(wsc)), vp(repack_bias(bsc)), vp(ws), ws.numel(), vp(y))
        _lib.check(rc)
        torch.cuda.synchronize()
    finally:
        L.gyre_debug_set_splitk_workspace(None, 0)
    report(f"conv3x3 + folded shortcut B{B} {H}x{H} {Cin}->{Cout} sc {C1}+{C2}", y.float().cpu().permute(0, 3, 1, 2), ref, TOL)
    # a shape the planner gives to another kernel is refused, not computed wrongly
    rc = L.gyre_op_conv3x3_shortcut(st(), vp(to_dev_bf16(nhwc(h[:1]))), 1, H, H, Cin, vp(repack_conv(w)), Cout, None,
                                    vp(to_dev_bf16(nhwc(x1[:1]))), C1, None, 0, vp(repack_linear(wsc[:, :C1])), None, vp(ws), ws.numel(), vp(y))
    assert rc in (0, -6)
