"""Dev tool: one conv shape under (cfg | ablation bits << 16) variants, interleaved rounds, median TFLOP/s.
usage: conv_ablate.py B H W Cin Cout variant [variant ...]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, W, Ci, Co = [int(a) for a in sys.argv[1:6]]
variants = [int(a, 0) for a in sys.argv[6:]]
rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
x, w, b = rnd(B, H, W, Ci), rnd(Co, 9 * Ci), torch.zeros(Co, device=DEV)
y = torch.empty(B, H, W, Co, dtype=torch.bfloat16, device=DEV)
ws = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
L.gyre_debug_set_splitk_workspace(vp(ws), ws.numel())
fl = 2.0 * B * H * W * Co * 9 * Ci
def run(): return L.gyre_op_conv3x3(st(), vp(x), B, H, W, Ci, vp(w), Co, vp(b), None, 1, 0, 0, vp(y))
def timeit(iters=10):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
res = {v: [] for v in variants}
for r in range(5):
    for v in variants:
        L.gyre_debug_force_gemm_cfg(v & 0xffff); L.gyre_debug_gemm_ablation(v >> 16)
        if run() != 0: res[v].append(float("nan")); continue
        res[v].append(timeit())
L.gyre_debug_force_gemm_cfg(0); L.gyre_debug_gemm_ablation(0)
print(f"conv {B}x{H}x{W} {Ci}->{Co}: " + " | ".join(f"cfg{v & 0xff}s{(v >> 8) & 0xff}/{v >> 16:#x}: {fl / statistics.median(res[v]) / 1e6:5.0f} TF ({statistics.median(res[v]):.1f} us)" for v in variants))
