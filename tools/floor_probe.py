"""Round 5: where do the ~25 us of a mid-size k_gemm8 launch go when loads, MFMAs and the epilogue are each switched off?  Times the
128x160 tile (config 8) on M x 1280 x 1280 for several M (grid size) with every ablation combination, warm."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, randn, repack_bias, repack_linear, st, vp
L = _lib.lib()
def timeit(fn, reps=15):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        _lib.prof_enable(None); fn(); torch.cuda.synchronize()
        c = _lib.prof_collect(); _lib.prof_enable([])
        ts.append(sum(v["ms"] for v in c.values()) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
def wall(fn, n=200):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
K = N = 1280
for cfg in (8,):
    for M in (128, 1024, 4096, 16384):
        x = randn(M, K, seed=1).to(torch.bfloat16).to(DEV)
        w, b = repack_linear(randn(N, K, seed=2) / math.sqrt(K)), repack_bias(randn(N, seed=3))
        y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        run = lambda: _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), None, 0, vp(y)))
        L.gyre_debug_force_gemm_cfg(cfg)
        out = []
        for bits in (0x9000000, 0x9000001, 0x1000000, 0x1000001, 0x8000000, 0):
            L.gyre_debug_gemm_ablation(bits)
            out.append(f"{bits:#x}:{timeit(run):6.1f}/{wall(run):6.1f}")
        L.gyre_debug_gemm_ablation(0); L.gyre_debug_force_gemm_cfg(0)
        print(f"cfg {cfg} M={M:6d}: event us / back-to-back wall us per launch by ablation bits (bit 27 = compiler-scheduled fragment reads; bit 24 = two-stage loop; 1 = no loads) " + "  ".join(out), flush=True)
