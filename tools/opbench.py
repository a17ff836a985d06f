"""Per-shape microbenchmarks of the hot kernels at the real SD1.5 (batch 16) shapes, through the C ABI.
usage: python tools/opbench.py [linear|conv|attn|gn|all] [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib

DEV = "cuda:0"
L = _lib.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def rnd(*shape):
    return (torch.randn(*shape, device=DEV) * 0.5).to(torch.bfloat16)


B = 16
_ws = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
L.gyre_debug_set_splitk_workspace(vp(_ws), _ws.numel())
if which in ("linear", "all"):
    print("== linear (M, K, N) ==")
    shapes = [(65536, 320, 320), (65536, 320, 640), (65536, 320, 2560), (65536, 1280, 320),
              (16384, 640, 640), (16384, 640, 1280), (16384, 640, 5120), (16384, 2560, 640),
              (4096, 1280, 1280), (4096, 1280, 2560), (4096, 1280, 10240), (4096, 5120, 1280),
              (1024, 1280, 1280), (1232, 768, 320), (1232, 768, 1280), (8192, 8192, 8192)]
    for M, K, N in shapes:
        geglu = N in (2560, 5120, 10240)
        x, w = rnd(M, K), rnd(N, K)
        bias = torch.zeros(N, device=DEV)
        nout = N // 2 if geglu else N
        y = torch.empty(M, nout, dtype=torch.bfloat16, device=DEV)
        res = None if geglu else rnd(M, nout)
        us = timeit(lambda: L.gyre_op_linear(st(), vp(x), M, K, vp(w), nout, vp(bias), vp(res), int(geglu), vp(y)))
        print(f"linear M{M:6d} K{K:5d} N{N:6d} geglu={int(geglu)}: {us:9.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s")

if which in ("conv", "all"):
    print("== conv3x3 (B,H,W,Cin,Cout,stride,ups) ==")
    shapes = [(B, 64, 64, 320, 320, 1, 0), (B, 64, 64, 640, 320, 1, 0), (B, 64, 64, 960, 320, 1, 0),
              (B, 64, 64, 320, 320, 2, 0), (B, 32, 32, 320, 640, 1, 0), (B, 32, 32, 640, 640, 1, 0),
              (B, 32, 32, 1280, 640, 1, 0), (B, 32, 32, 1920, 640, 1, 0), (B, 32, 32, 640, 640, 1, 1),
              (B, 16, 16, 640, 1280, 1, 0), (B, 16, 16, 1280, 1280, 1, 0), (B, 16, 16, 2560, 1280, 1, 0),
              (B, 8, 8, 1280, 1280, 1, 0), (B, 8, 8, 2560, 1280, 1, 0),
              (8, 64, 64, 512, 512, 1, 0), (8, 128, 128, 512, 512, 1, 0), (8, 256, 256, 256, 256, 1, 0),
              (8, 512, 512, 128, 128, 1, 0), (8, 256, 256, 256, 256, 1, 1)]
    for Bc, H, W, Ci, Co, s, ups in shapes:
        x, w = rnd(Bc, H, W, Ci), rnd(Co, 9 * Ci)
        bias = torch.zeros(Co, device=DEV)
        Ho, Wo = (H * (2 if ups else 1)) // s, (W * (2 if ups else 1)) // s
        y = torch.empty(Bc, Ho, Wo, Co, dtype=torch.bfloat16, device=DEV)
        us = timeit(lambda: L.gyre_op_conv3x3(st(), vp(x), Bc, H, W, Ci, vp(w), Co, vp(bias), None, s, ups, 0, vp(y)),
                    max(3, iters // 2))
        fl = 2.0 * Bc * Ho * Wo * Co * 9 * Ci
        print(f"conv B{Bc} {H}x{W} {Ci:4d}->{Co:4d} s{s} ups{ups}: {us:9.1f} us  {fl/us/1e6:7.1f} TF/s")

if which in ("attn", "all"):
    print("== attention (B,heads,Nq,Nk,D) ==")
    for Bc, h, Nq, Nk, D in [(B, 8, 4096, 4096, 40), (B, 8, 1024, 1024, 80), (B, 8, 256, 256, 160), (B, 8, 64, 64, 160),
                             (B, 8, 4096, 77, 40), (B, 8, 1024, 77, 80), (8, 1, 4096, 4096, 512)]:
        Cc = h * D
        q, k = rnd(Bc, Nq, Cc), rnd(Bc, Nk, Cc)
        ldvt = (Nk + 7) // 8 * 8
        vt = rnd(Bc, Cc, ldvt)
        o = torch.empty(Bc, Nq, Cc, dtype=torch.bfloat16, device=DEV)
        fl = 4.0 * Bc * h * Nq * Nk * D
        res = []
        for var in (1, 2, 4):
            L.gyre_debug_force_attn_variant(var)
            us = timeit(lambda: L.gyre_op_attention(st(), vp(q), Cc, vp(k), Cc, vp(vt), ldvt, Bc, h, Nq, Nk, D, vp(o), Cc))
            res.append(f"v{var}: {us:8.1f} us {fl/us/1e6:6.1f} TF/s")
        L.gyre_debug_force_attn_variant(0)
        print(f"attn B{Bc} h{h} Nq{Nq} Nk{Nk} D{D}: " + " | ".join(res))

if which in ("gn", "all"):
    print("== groupnorm+silu / layernorm ==")
    for Bc, HW, Cc in [(B, 4096, 320), (B, 4096, 960), (B, 1024, 640), (B, 256, 1280), (B, 64, 2560), (8, 262144, 128)]:
        x = rnd(Bc, HW, Cc)
        y = torch.empty_like(x)
        g, b = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
        wsb = L.gyre_op_groupnorm_workspace(Bc, HW, Cc, 32)
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        us = timeit(lambda: L.gyre_op_groupnorm(st(), vp(x), None, 0, Bc, HW, Cc, 32, vp(g), vp(b), 1e-5, 1, vp(ws), wsb, vp(y)))
        print(f"groupnorm B{Bc} HW{HW} C{Cc}: {us:8.1f} us  {Bc*HW*Cc*6/us/1e3:7.1f} GB/s (read x2 + write)")
    for M, Cc in [(65536, 320), (16384, 640), (4096, 1280)]:
        x = rnd(M, Cc)
        y = torch.empty_like(x)
        g, b = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
        us = timeit(lambda: L.gyre_op_layernorm(st(), vp(x), M, Cc, vp(g), vp(b), 1e-5, vp(y)))
        print(f"layernorm M{M} C{Cc}: {us:8.1f} us  {M*Cc*4/us/1e3:7.1f} GB/s")
    src = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
    dst = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
    us = timeit(lambda: L.gyre_op_copy_probe(st(), vp(src), vp(dst), 1 << 30), 5)
    print(f"copy probe 1 GiB: {us:.1f} us  {2*(1<<30)/us/1e3:.1f} GB/s (read+write)")
