"""Dev tool: plain square GEMMs (uniform random operands) under given tile configs - comparable with the guide's tables."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
cfgs = [int(a, 0) for a in sys.argv[1:]] or [6, 13]   # cfg | ablation_bits << 16
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for n in (4096, 8192, (65536, 2880, 320), (65536, 8640, 320), (16384, 5760, 640), (65536, 320, 2560), (16384, 640, 5120)):
    M_, K_, N_ = (n, n, n) if isinstance(n, int) else n
    x = (torch.rand(M_, K_, device=DEV) * 2 - 1).to(torch.bfloat16); w = (torch.rand(N_, K_, device=DEV) * 2 - 1).to(torch.bfloat16)
    y = torch.empty(M_, N_, dtype=torch.bfloat16, device=DEV)
    ref = None
    r = []
    for c in cfgs:
        L.gyre_debug_force_gemm_cfg(c & 0xffff); L.gyre_debug_gemm_ablation(c >> 16)
        rc = L.gyre_op_linear(st(), vp(x), M_, K_, vp(w), N_, None, None, 0, vp(y))
        if rc: r.append(f"cfg{c}: n/a"); continue
        torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        same = bool(torch.equal(ref, y)) or f"{float((y.float()-ref.float()).norm()/ref.float().norm()):.1e}"
        us = timeit(lambda: L.gyre_op_linear(st(), vp(x), M_, K_, vp(w), N_, None, None, 0, vp(y)))
        r.append(f"cfg{c&0xffff}/{c>>16:#x}: {2.0*M_*K_*N_/us/1e6:6.0f} TF/s same={same}")
    print(f"{M_}x{K_}x{N_}: " + " | ".join(r))
L.gyre_debug_force_gemm_cfg(0); L.gyre_debug_gemm_ablation(0)
