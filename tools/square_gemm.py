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
for n in (2560, 4096, 5120, 8192):
    x = (torch.rand(n, n, device=DEV) * 2 - 1).to(torch.bfloat16); w = (torch.rand(n, n, device=DEV) * 2 - 1).to(torch.bfloat16)
    y = torch.empty(n, n, dtype=torch.bfloat16, device=DEV)
    ref = None
    r = []
    for c in cfgs:
        L.gyre_debug_force_gemm_cfg(c & 0xffff); L.gyre_debug_gemm_ablation(c >> 16)
        rc = L.gyre_op_linear(st(), vp(x), n, n, vp(w), n, None, None, 0, vp(y))
        if rc: r.append(f"cfg{c}: n/a"); continue
        torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        same = bool(torch.equal(ref, y))
        us = timeit(lambda: L.gyre_op_linear(st(), vp(x), n, n, vp(w), n, None, None, 0, vp(y)))
        r.append(f"cfg{c&0xffff}/{c>>16:#x}: {2.0*n**3/us/1e6:6.0f} TF/s same={same}")
    print(f"{n}^3: " + " | ".join(r))
L.gyre_debug_force_gemm_cfg(0); L.gyre_debug_gemm_ablation(0)
