"""Small-problem GEMM kernel (tile config 32) against the register-staged 4-wave kernel (tuning bit 5) on the linear shapes of the
deep UNet levels at small batch: kernel time per launch from the library's HIP events, warm and (COLD=1) behind a cache-evicting copy."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, randn, repack_bias, repack_linear, st, vp
L = _lib.lib()
COLD = os.environ.get("COLD", "0") == "1"
ev_a = torch.empty(160 << 20, dtype=torch.uint8, device=DEV); ev_b = torch.empty(160 << 20, dtype=torch.uint8, device=DEV)
FORCE = [int(c, 0) for c in os.environ.get("FORCE", "").split(",") if c]        # extra forced configs (cfg | splits << 8)
wsk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
L.gyre_debug_set_splitk_workspace(vp(wsk), wsk.numel())
def timeit(fn, reps=15):
    ts = []
    for _ in range(reps):
        if COLD: ev_b.copy_(ev_a)
        torch.cuda.synchronize()
        _lib.prof_enable(None); fn(); torch.cuda.synchronize()
        c = _lib.prof_collect(); _lib.prof_enable([])
        ts.append(sum(v["ms"] for v in c.values()) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for (M, K, N, res) in [(2048, 640, 640, 1), (2048, 640, 640, 0), (512, 1280, 1280, 1), (512, 1280, 1280, 0), (128, 1280, 1280, 1), (154, 768, 320, 0),
                       (154, 768, 1280, 0), (2048, 1280, 1280, 1), (1024, 640, 640, 1), (4096, 1280, 1280, 1), (8192, 640, 640, 1), (8192, 320, 320, 1),
                       (2048, 2560, 640, 1), (512, 5120, 1280, 1)]:
    x = (randn(M, K, seed=1)).to(torch.bfloat16).to(DEV)
    w, b = repack_linear(randn(N, K, seed=2) / math.sqrt(K)), repack_bias(randn(N, seed=3))
    r = randn(M, N, seed=4).to(torch.bfloat16).to(DEV) if res else None
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    run = lambda: _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), vp(r), 0, vp(y)))
    out = []
    for bits, force in [(0x20, 0), (0, 0)] + [(0, f) for f in FORCE]:
        L.gyre_debug_gemm_ablation(bits); L.gyre_debug_force_gemm_cfg(force)
        try:
            run(); torch.cuda.synchronize()
            _lib.prof_enable(None); run(); torch.cuda.synchronize(); names = list(_lib.prof_collect()); _lib.prof_enable([])
            out.append((timeit(run), names))
        except Exception as e:
            out.append((float("nan"), [str(e)[:30]]))
    L.gyre_debug_gemm_ablation(0); L.gyre_debug_force_gemm_cfg(0)
    fl = 2.0 * M * N * K
    print(f"M={M:5d} K={K:5d} N={N:5d} res={res}: " + " | ".join(f"{t:6.1f} us {n[0][:22] if n else ''}" for t, n in out), flush=True)
