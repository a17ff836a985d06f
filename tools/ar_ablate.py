"""Where the A-resident kernel's time goes (64x64 GEGLU FF1 shape): timing ablations compiled in with GYRE_AR_ABLATIONS
(touch gyre_amd/csrc/kernels_gemm_ar.hip && GYRE_AR_ABLATIONS=1 python -c 'from gyre_amd import build as b; b.build()').
Results of the ablated runs are garbage, their times are what matters."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, randn, repack_bias, repack_linear, st, vp
L = _lib.lib()
M, K, F_ = int(os.environ.get("M", 65536)), 320, 1280
x = (randn(M, K, seed=1) * 1.3).to(torch.bfloat16).to(DEV)
w = repack_linear(randn(2 * F_, K, seed=2) / math.sqrt(K), geglu=True)
b = repack_bias(randn(2 * F_, seed=3) * 0.3, geglu=True)
y = torch.empty(M, F_, dtype=torch.bfloat16, device=DEV)
arws = torch.empty(5120 * 640 * 2, dtype=torch.uint8, device=DEV)
ev_a = torch.empty(160 << 20, dtype=torch.uint8, device=DEV); ev_b = torch.empty_like(ev_a)
L.gyre_debug_set_ar_workspace(vp(arws), arws.numel())
run = lambda: _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), F_, vp(b), None, 1, vp(y)))
names = {0: "full", 1: "no epilogue arithmetic", 2: "no MFMA", 3: "no epilogue, no MFMA", 4: "no stores", 5: "no epilogue, no stores",
         8: "no fragment reads", 16: "no ring requests / waits", 7: "no epilogue / MFMA / stores", 10: "no MFMA, no fragment reads",
         13: "MFMA only (no epilogue / reads / stores)", 12: "MFMA + epilogue (no reads / stores)",
         32: "V: read pinned before MFMA", 64: "V: prefetch 10", 96: "V: read first + prefetch 10", 128: "V: setprio 1 on waves 4-7",
         256: "V: epilogue stage before MFMA", 160: "V: read first + setprio", 37: "no epi/stores, read first", 69: "no epi/stores, prefetch 10"}
for cold in (False, True):
    for abl in ((0, 1, 2, 3, 4, 5, 8, 16, 7, 10, 13, 12, 0, 32, 64, 96, 128, 256, 160, 37, 69, 0) if not cold else (0, 32, 64, 96, 128, 256, 160, 0)):
        L.gyre_debug_gemm_ablation(abl << 23)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            if cold: ev_b.copy_(ev_a)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); e.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(e) * 1e3)
        ts.sort()
        print(f"{'cold' if cold else 'warm'} abl {abl:2d} {names[abl]:32s} {ts[len(ts) // 2]:7.1f} us (min {ts[0]:.1f})", flush=True)
L.gyre_debug_gemm_ablation(0)
