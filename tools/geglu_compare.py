"""Dev tool: GEGLU FF1 shapes of SD1.5 (batch 16) under different tile configs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
cfgs = [int(a, 0) for a in sys.argv[1:]] or [0, 6, 7, 10]
COLD = os.environ.get("COLD") == "1"      # evict L2 / Infinity Cache before every timed launch (the in-UNet regime)
if COLD:
    fl_a = torch.empty(384 << 20, dtype=torch.uint8, device=DEV); fl_b = torch.empty(384 << 20, dtype=torch.uint8, device=DEV)
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    if COLD:
        tot = 0.0
        for _ in range(6):
            fl_a.copy_(fl_b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / 6 * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, K, F) in [(65536, 320, 1280), (16384, 640, 2560), (4096, 1280, 5120)]:
    x, w, b = rnd(M, K), rnd(2 * F, K), torch.zeros(2 * F, device=DEV)
    y = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    fl = 2.0 * M * 2 * F * K
    r = []
    for c in cfgs:
        L.gyre_debug_force_gemm_cfg(c)
        rc = L.gyre_op_linear(st(), vp(x), M, K, vp(w), F, vp(b), None, 1, vp(y))
        if rc: r.append(f"cfg{c}: n/a"); continue
        us = timeit(lambda: L.gyre_op_linear(st(), vp(x), M, K, vp(w), F, vp(b), None, 1, vp(y)))
        r.append(f"cfg{c}: {us:7.1f}us {fl/us/1e6:5.0f}TF")
    print(f"geglu {M}x{K}->2x{F}: " + " | ".join(r))
    # FF2
    x2, w2 = rnd(M, F), rnd(K, F)
    y2 = torch.empty(M, K, dtype=torch.bfloat16, device=DEV); rs = rnd(M, K)
    fl = 2.0 * M * K * F
    r = []
    for c in cfgs:
        L.gyre_debug_force_gemm_cfg(c)
        rc = L.gyre_op_linear(st(), vp(x2), M, F, vp(w2), K, vp(b[:K].contiguous()), vp(rs), 0, vp(y2))
        if rc: r.append(f"cfg{c}: n/a"); continue
        us = timeit(lambda: L.gyre_op_linear(st(), vp(x2), M, F, vp(w2), K, vp(b[:K].contiguous()), vp(rs), 0, vp(y2)))
        r.append(f"cfg{c}: {us:7.1f}us {fl/us/1e6:5.0f}TF")
    print(f"ff2   {M}x{F}->{K}: " + " | ".join(r))
L.gyre_debug_force_gemm_cfg(0)
