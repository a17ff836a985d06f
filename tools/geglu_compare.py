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
COLD = os.environ.get("COLD") in ("1", "2")   # 1: evict L2 / Infinity Cache before every timed launch (the in-UNet regime)
POLLUTE = os.environ.get("COLD") == "2"         # 2: ... and run four other full-chip kernels first (instruction caches hold THEIR code)
if COLD:
    fl_a = torch.empty(384 << 20, dtype=torch.uint8, device=DEV); fl_b = torch.empty(384 << 20, dtype=torch.uint8, device=DEV)
if POLLUTE:
    px = rnd(16, 64, 64, 320); pw = rnd(320, 9 * 320); pb = torch.zeros(1280, device=DEV); py = torch.empty(16, 64, 64, 320, dtype=torch.bfloat16, device=DEV)
    pl = rnd(65536, 320); plw = rnd(320, 320); ply = torch.empty(65536, 320, dtype=torch.bfloat16, device=DEV)
    pq = rnd(2, 4096, 320); pvt = rnd(2, 320, 4096); po = torch.empty(2, 4096, 320, dtype=torch.bfloat16, device=DEV)
    pw2 = rnd(1280, 9 * 320); py2 = torch.empty(16, 32, 32, 1280, dtype=torch.bfloat16, device=DEV); px2 = rnd(16, 32, 32, 320)
    def pollute():
        force = L.gyre_debug_force_gemm_cfg(0)
        L.gyre_op_conv3x3(st(), vp(px), 16, 64, 64, 320, vp(pw), 320, vp(pb), None, 1, 0, 0, vp(py))
        L.gyre_op_attention_ex(st(), vp(pq), 320, vp(pq), 320, vp(pvt), 4096, 2, 8, 4096, 4096, 40, vp(po), 320, 1)
        L.gyre_op_linear(st(), vp(pl), 65536, 320, vp(plw), 320, vp(pb), vp(pl), 0, vp(ply))
        L.gyre_op_conv3x3(st(), vp(px2), 16, 32, 32, 320, vp(pw2), 1280, vp(pb), None, 1, 0, 0, vp(py2))
        L.gyre_debug_force_gemm_cfg(force)
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    if COLD:
        tot = 0.0
        for _ in range(6):
            if POLLUTE: pollute()
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
