"""Compare GEMM tile configurations on the SD1.5 (batch 16) shapes.  usage: cfg_compare.py [cfg ids...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
cfgs = [int(a, 0) for a in sys.argv[1:]] or [0, 4, 5, 9]   # cfg | splits << 8 | ablation bits << 16
ws = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
L.gyre_debug_set_splitk_workspace(vp(ws), ws.numel())
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = 16
print("cfgs:", cfgs, "(0 = planner)")
for (Bc, H, W, Ci, Co) in [(B,64,64,320,320),(B,64,64,960,320),(B,32,32,320,640),(B,32,32,640,640),(B,32,32,960,640),(B,32,32,1280,640),(B,32,32,1920,640),(B,16,16,640,1280),(B,16,16,1280,1280),(B,16,16,2560,1280),(B,8,8,1280,1280),(B,8,8,2560,1280)]:
    x, w, b = rnd(Bc, H, W, Ci), rnd(Co, 9 * Ci), torch.zeros(Co, device=DEV)
    y = torch.empty(Bc, H, W, Co, dtype=torch.bfloat16, device=DEV)
    fl = 2.0 * Bc * H * W * Co * 9 * Ci
    r = []
    for c in cfgs:
        L.gyre_debug_force_gemm_cfg(c & 0xffff); L.gyre_debug_gemm_ablation(c >> 16)
        rc = L.gyre_op_conv3x3(st(), vp(x), Bc, H, W, Ci, vp(w), Co, vp(b), None, 1, 0, 0, vp(y))
        r.append(f"cfg{c&255}s{(c>>8)&255}/{c>>16:#x}: {fl/timeit(lambda: L.gyre_op_conv3x3(st(), vp(x), Bc, H, W, Ci, vp(w), Co, vp(b), None, 1, 0, 0, vp(y)))/1e6:6.0f}" if rc == 0 else f"cfg{c&255}s{c>>8}:   n/a")
    print(f"conv {Bc}x{H}x{W} {Ci:4d}->{Co:4d}: " + " | ".join(r) + " TF/s")
for (M, K, N, res) in [(65536,320,320,1),(65536,320,640,0),(65536,1280,320,1),(16384,640,640,1),(16384,640,1280,0),(16384,2560,640,1),(4096,1280,1280,1),(4096,5120,1280,1)]:
    x, w, b = rnd(M, K), rnd(N, K), torch.zeros(N, device=DEV)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); rs = rnd(M, N) if res else None
    fl = 2.0 * M * N * K
    r = []
    for c in cfgs:
        L.gyre_debug_force_gemm_cfg(c)
        rc = L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), vp(rs), 0, vp(y))
        r.append(f"cfg{c&255}s{c>>8}: {timeit(lambda: L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), vp(rs), 0, vp(y))):6.1f}us" if rc == 0 else f"cfg{c&255}s{c>>8}:   n/a")
    print(f"linear {M}x{K}x{N} res={res}: " + " | ".join(r))
L.gyre_debug_force_gemm_cfg(0); L.gyre_debug_gemm_ablation(0)
