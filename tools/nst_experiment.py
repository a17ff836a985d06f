import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
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
for (M,K,N) in [(8192,8192,8192),(65536,1280,256),(16384,2560,1280)]:
    x, w, b = rnd(M, K), rnd(N, K), torch.zeros(N, device=DEV)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); y2=torch.empty_like(y)
    r={}
    for cfg in (7,8,6):
        L.gyre_debug_force_gemm_cfg(cfg)
        out = y if cfg==7 else y2
        f=lambda: L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), None, 0, vp(out))
        r[cfg]=timeit(f)
    L.gyre_debug_force_gemm_cfg(0)
    print(f"linear {M}x{K}x{N}: cfg7(128x256,2st) {r[7]:.1f} us {2*M*N*K/r[7]/1e6:.0f} TF | cfg8(3st) {r[8]:.1f} us {2*M*N*K/r[8]/1e6:.0f} TF | cfg6(256x256) {r[6]:.1f} us  equal={bool(torch.equal(y,y2))}")
for (B,H,W,Ci,Co) in [(16,32,32,1280,1280),(16,64,64,320,256)]:
    x, w, b = rnd(B, H, W, Ci), rnd(Co, 9 * Ci), torch.zeros(Co, device=DEV)
    y = torch.empty(B, H, W, Co, dtype=torch.bfloat16, device=DEV)
    r={}
    for cfg in (7,8):
        L.gyre_debug_force_gemm_cfg(cfg)
        f=lambda: L.gyre_op_conv3x3(st(), vp(x), B, H, W, Ci, vp(w), Co, vp(b), None, 1, 0, 0, vp(y))
        r[cfg]=timeit(f)
    L.gyre_debug_force_gemm_cfg(0)
    fl=2*B*H*W*Co*9*Ci
    print(f"conv {B}x{H}x{W} {Ci}->{Co}: cfg7 {r[7]:.1f} us {fl/r[7]/1e6:.0f} TF | cfg8(3st) {r[8]:.1f} us {fl/r[8]/1e6:.0f} TF")
L.gyre_debug_gemm_ablation.argtypes=[C.c_int]
print("-- loads-only (no MFMA) mode: is the DMA latency- or bandwidth-bound?")
for (M,K,N) in [(8192,8192,8192),(16384,2560,1280)]:
    x, w, b = rnd(M, K), rnd(N, K), torch.zeros(N, device=DEV)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    r={}
    for cfg in (7,8):
        L.gyre_debug_force_gemm_cfg(cfg)
        for bits in (0,2,1):
            L.gyre_debug_gemm_ablation(bits)
            r[(cfg,bits)]=timeit(lambda: L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), None, 0, vp(y)))
    L.gyre_debug_gemm_ablation(0); L.gyre_debug_force_gemm_cfg(0)
    nb=(M//128)*(N//256); steps=K//64
    print(f"linear {M}x{K}x{N}: 2-stage full {r[(7,0)]:.0f} loads-only {r[(7,2)]:.0f} mfma-only {r[(7,1)]:.0f} | 3-stage full {r[(8,0)]:.0f} loads-only {r[(8,2)]:.0f} mfma-only {r[(8,1)]:.0f} us;  bytes moved {nb*steps*49152/1e9:.2f} GB")
