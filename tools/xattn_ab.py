import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from gyre_amd import _lib
L = _lib.lib(); dev = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
fl_a = torch.empty(384 << 20, dtype=torch.uint8, device=dev); fl_b = torch.empty(384 << 20, dtype=torch.uint8, device=dev)
def timeit(fn, cold):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(8):
        if cold: fl_a.copy_(fl_b)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); tot += a.elapsed_time(b)
    return tot / 8 * 1e3
for (B, H, Nq, D) in [(16, 8, 4096, 40), (16, 8, 1024, 80), (16, 8, 256, 160), (16, 8, 64, 160), (2, 8, 4096, 40), (2, 10, 4096, 64)]:
    Cc = H * D; Nk = 77; ldvt = 80
    q = torch.randn(B, Nq, Cc, device=dev).to(torch.bfloat16)
    k = (torch.randn(B, Nk, Cc, device=dev) * 0.3).to(torch.bfloat16)
    vt = torch.randn(B, Cc, ldvt, device=dev).to(torch.bfloat16)
    outs = {}
    for var in (6, 0):
        o = torch.zeros(B, Nq, Cc, dtype=torch.bfloat16, device=dev)
        L.gyre_debug_force_attn_variant(var)
        fn = lambda: L.gyre_op_attention_ex(st(), vp(q), Cc, vp(k), Cc, vp(vt), ldvt, B, H, Nq, Nk, D, vp(o), Cc, 1)
        rc = fn(); assert rc == 0, L.gyre_last_error()
        outs[var] = (o.clone(), timeit(fn, False), timeit(fn, True))
    L.gyre_debug_force_attn_variant(0)
    same = torch.equal(outs[0][0], outs[6][0])
    print(f"B{B} H{H} Nq{Nq} D{D}: one block per WG warm {outs[6][1]:.1f} cold {outs[6][2]:.1f} us | looped warm {outs[0][1]:.1f} cold {outs[0][2]:.1f} us | bits equal {same}")
