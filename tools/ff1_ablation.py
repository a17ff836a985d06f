import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, K, F) in [(65536, 320, 1280), (16384, 640, 2560)]:
    x, w, b = rnd(M, K), rnd(2 * F, K), torch.zeros(2 * F, device=DEV)
    y = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    yp = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=DEV)
    r = []
    for bits in (0, 4, 8, 16, 24, 1, 2, 3):
        L.gyre_debug_gemm_ablation(bits)
        r.append(f"abl{bits}: {timeit(lambda: L.gyre_op_linear(st(), vp(x), M, K, vp(w), F, vp(b), None, 1, vp(y))):6.1f}us")
    L.gyre_debug_gemm_ablation(0)
    # same GEMM without the GEGLU epilogue (plain bias epilogue, twice the output)
    r.append(f"plain 2F out: {timeit(lambda: L.gyre_op_linear(st(), vp(x), M, K, vp(w), 2 * F, vp(b), None, 0, vp(yp))):6.1f}us")
    print(f"FF1 {M}x{K}->2x{F}: " + " | ".join(r))
# the form the UNet runs: LayerNorm folded into the GEMM (gyre_op_ln_linear); GEMM kernel time alone from the library's own
# per-class HIP events.  bit 19 (0x80000) = the slab-outer epilogue (before the pair-outer form), 8 = no GELU, 16 = no stores
for (M, K, F) in [(65536, 320, 1280), (16384, 640, 2560)]:
    x, w, b = rnd(M, K), rnd(2 * F, K), torch.zeros(2 * F, device=DEV)
    gam, bet = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    y = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    wsb = L.gyre_op_ln_linear_workspace(2 * F, K, M)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    r = []
    for bits in (0, 0x80000, 8, 0x80008, 16, 24, 4):
        L.gyre_debug_gemm_ablation(bits)
        fn = lambda: L.gyre_op_ln_linear(st(), vp(x), M, K, vp(gam), vp(bet), C.c_float(1e-5), vp(w), F, vp(b), 1, 0, None, 0, None, 0, vp(ws), wsb, vp(y))
        for _ in range(3): rc = fn()
        torch.cuda.synchronize()
        _lib.prof_enable(["k_gemm8<", "k_gemm4s<"]); _lib.prof_collect()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        pr = _lib.prof_collect(); _lib.prof_enable([])
        us = sum(v["ms"] for v in pr.values()) / 10 * 1e3
        r.append(f"{bits:#x}: rc={rc} {us:6.1f}us")
    L.gyre_debug_gemm_ablation(0)
    print(f"LN-folded FF1 {M}x{K}->2x{F} (GEMM kernel only): " + " | ".join(r))
