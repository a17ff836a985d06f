"""A-resident GEMM kernel (tile config 30) against the planner's previous choice on the K = 320 / 640 layer shapes of the SD1.5
UNet at batch 16 (and batch 2): time per launch (HIP events, cache-evicting copy between launches when COLD=1)."""
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
evict_a = torch.empty(160 << 20, dtype=torch.uint8, device=DEV)
evict_b = torch.empty(160 << 20, dtype=torch.uint8, device=DEV)
arws = torch.empty(5120 * 640 * 2, dtype=torch.uint8, device=DEV)


def timeit(fn, reps=12):
    ts = []
    for _ in range(reps):
        if COLD:
            evict_b.copy_(evict_a)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def case(name, M, K, N, geglu=False, res=False, ln=False):
    rows = 2 * N if geglu else N
    x = (randn(M, K, seed=1) * 1.3).to(torch.bfloat16).to(DEV)
    w = repack_linear(randn(rows, K, seed=2) / math.sqrt(K), geglu=geglu)
    b = repack_bias(randn(rows, seed=3) * 0.3, geglu=geglu)
    r = randn(M, N, seed=4).to(torch.bfloat16).to(DEV) if res else None
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    g, be = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    ws = torch.empty(L.gyre_op_ln_linear_workspace(rows, K, M), dtype=torch.uint8, device=DEV) if ln else None

    def run():
        if ln:
            _lib.check(L.gyre_op_ln_linear(st(), vp(x), M, K, vp(g), vp(be), 1e-5, vp(w), N, vp(b), int(geglu), 0, None, 0, None, 0,
                                           vp(ws), ws.numel(), vp(y)))
        else:
            _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), vp(r), int(geglu), vp(y)))
    out = []
    for on in (False, True):
        torch.cuda.synchronize()
        L.gyre_debug_set_ar_workspace(vp(arws) if on else None, arws.numel() if on else 0)
        L.gyre_debug_gemm_ablation(0x400000 if on else 0)
        run(); torch.cuda.synchronize()
        out.append(timeit(run))
    L.gyre_debug_set_ar_workspace(None, 0)
    L.gyre_debug_gemm_ablation(0)
    fl = 2.0 * M * rows * K
    print(f"{name:34s} M={M:6d} K={K:4d} N={rows:5d}  tiles {out[0]:7.1f} us ({fl / out[0] / 1e6:6.0f} TF/s)   "
          f"a-resident {out[1]:7.1f} us ({fl / out[1] / 1e6:6.0f} TF/s)   x{out[0] / out[1]:.2f}", flush=True)


# (the LayerNorm-folded forms include the fold + statistics launches of the bare operator: compare the plain forms for kernel time)
for B in (16, 2):
    print(f"--- batch {B} ({'cold' if COLD else 'warm'})")
    case("64x64 GEGLU FF1", B * 4096, 320, 1280, geglu=True)
    case("64x64 to_q / proj_in", B * 4096, 320, 320)
    case("64x64 to_out + residual", B * 4096, 320, 320, res=True)
    case("64x64 Q|K|V as plain N=960", B * 4096, 320, 960)
    case("32x32 GEGLU FF1", B * 1024, 640, 2560, geglu=True)
