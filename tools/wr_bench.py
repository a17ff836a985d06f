"""W-resident GEMM kernel (tile config 31) against the planner's previous choice on the square projections of the SD1.5 UNet
(K = N = 320 at 64x64, 640 at 32x32): kernel time per launch from the library's own HIP events (GYRE profiling classes; the
on-the-fly weight packing of the bare operator is not counted), cache-evicting copy between launches when COLD=1, and the
largest difference between the two kernels' outputs."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, randn, repack_bias, repack_linear, st, vp

L = _lib.lib()
COLD = os.environ.get("COLD", "1") == "1"
ABL = int(os.environ.get("ABL", "0"))
evict_a = torch.empty(160 << 20, dtype=torch.uint8, device=DEV)
evict_b = torch.empty(160 << 20, dtype=torch.uint8, device=DEV)
arws = torch.empty(5120 * 640 * 2, dtype=torch.uint8, device=DEV)


def timeit(fn, reps=12):
    ts = []
    for _ in range(reps):
        if COLD:
            evict_b.copy_(evict_a)
        torch.cuda.synchronize()
        _lib.prof_enable(["k_gemm", "k_g8", "k_g4s"])
        fn()
        torch.cuda.synchronize()
        c = _lib.prof_collect()
        _lib.prof_enable([])
        ts.append(sum(v["ms"] for v in c.values()) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def case(name, M, K, N, res=False, ln=False):
    x = (randn(M, K, seed=1) * 1.3).to(torch.bfloat16).to(DEV)
    w = repack_linear(randn(N, K, seed=2) / math.sqrt(K))
    b = repack_bias(randn(N, seed=3) * 0.3)
    r = randn(M, N, seed=4).to(torch.bfloat16).to(DEV) if res else None
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    g, be = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(N, K, M), dtype=torch.uint8, device=DEV) if ln else None

    def run():
        if ln:
            _lib.check(L.gyre_op_ln_linear(st(), vp(x), M, K, vp(g), vp(be), 1e-5, vp(w), N, vp(b), 0, 0, None, 0, None, 0,
                                           vp(ws), ws.numel(), vp(y)))
        else:
            _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), vp(r), 0, vp(y)))
    out, ys = [], []
    for on in (False, True):
        torch.cuda.synchronize()
        L.gyre_debug_set_ar_workspace(vp(arws) if on else None, arws.numel() if on else 0)
        L.gyre_debug_gemm_ablation(((ABL << 23) | 0x2000) if on else 0)
        run(); torch.cuda.synchronize()
        ys.append(y.float().clone())
        out.append(timeit(run))
    L.gyre_debug_set_ar_workspace(None, 0)
    L.gyre_debug_gemm_ablation(0)
    fl = 2.0 * M * N * K
    by = M * K * 2 + M * N * 2 * (2 if res else 1)
    d = (ys[0] - ys[1]).abs().max().item()
    print(f"{name:30s} M={M:6d} K={K:4d} N={N:5d}  tiles {out[0]:6.1f} us   w-resident {out[1]:6.1f} us ({fl / out[1] / 1e6:5.0f} TF/s, "
          f"{by / out[1] / 1e6:4.2f} TB/s)   x{out[0] / out[1]:.2f}   max |diff| {d:.3g} (|y| max {ys[0].abs().max().item():.3g})", flush=True)


for B in ((16, 2) if not ABL else (16,)):
    print(f"--- batch {B} ({'cold' if COLD else 'warm'})")
    case("64x64 proj_in", B * 4096, 320, 320)
    case("64x64 to_out + residual", B * 4096, 320, 320, res=True)
    if os.environ.get("LN", "1") == "1":
        case("64x64 to_q (folded LayerNorm)", B * 4096, 320, 320, ln=True)
    if os.environ.get("K640", "0") == "1":
        case("32x32 proj_in", B * 1024, 640, 640)
        case("32x32 to_out + residual", B * 1024, 640, 640, res=True)
