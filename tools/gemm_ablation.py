import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
L.gyre_debug_gemm_ablation.argtypes=[C.c_int]; L.gyre_debug_gemm_ablation.restype=C.c_int
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B,H,W,Ci,Co) in [(16,64,64,320,320),(16,32,32,1280,640)]:
    x, w, b = rnd(B, H, W, Ci), rnd(Co, 9 * Ci), torch.zeros(Co, device=DEV)
    y = torch.empty(B, H, W, Co, dtype=torch.bfloat16, device=DEV)
    f=lambda: L.gyre_op_conv3x3(st(), vp(x), B, H, W, Ci, vp(w), Co, vp(b), None, 1, 0, 0, vp(y))
    r=[]
    for bits in (0,1,2,3,7,4):
        L.gyre_debug_gemm_ablation(bits); r.append(timeit(f))
    L.gyre_debug_gemm_ablation(0)
    print(f"conv {B}x{H}x{W} {Ci}->{Co}: full {r[0]:.1f} us | no-loads {r[1]:.1f} | no-mfma {r[2]:.1f} | neither {r[3]:.1f} | neither+no-epilogue {r[4]:.1f} | full-no-epilogue {r[5]:.1f}")
for (M,K,N,gg) in [(8192,8192,8192,0),(65536,1280,320,0),(65536,320,2560,1),(65536,320,640,0),(16384,640,5120,1),(16384,640,1280,0)]:
    x, w, b = rnd(M, K), rnd(N, K), torch.zeros(N, device=DEV)
    no = N // 2 if gg else N
    y = torch.empty(M, no, dtype=torch.bfloat16, device=DEV)
    f=lambda: L.gyre_op_linear(st(), vp(x), M, K, vp(w), no, vp(b), None, gg, vp(y))
    r=[]
    for bits in (0,1,2,3,7,4):
        L.gyre_debug_gemm_ablation(bits); r.append(timeit(f))
    L.gyre_debug_gemm_ablation(0)
    print(f"linear {M}x{K}x{N} geglu={gg}: full {r[0]:.1f} us | no-loads {r[1]:.1f} | no-mfma {r[2]:.1f} | neither {r[3]:.1f} | neither+no-epilogue {r[4]:.1f} | full-no-epilogue {r[5]:.1f}")
