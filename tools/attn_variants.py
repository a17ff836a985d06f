"""Dev tool: accuracy + time of the attention kernel variants on the SD1.5 (batch 16) shapes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
variants = [int(a) for a in sys.argv[1:]] or [0, 3]
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, h, Nq, Nk, D, scale) in [(16, 8, 4096, 4096, 40, 1.0), (16, 8, 1024, 1024, 80, 1.0), (16, 8, 256, 256, 160, 1.0),
                                 (16, 8, 4096, 77, 40, 1.0), (4, 8, 4096, 4096, 40, 6.0), (2, 8, 1000, 1003, 40, 1.0),
                                 (4, 8, 9216, 9216, 40, 1.0), (2, 10, 4096, 4096, 64, 1.0), (2, 8, 2304, 2304, 80, 1.0)]:
    Cc = h * D
    g = torch.Generator(device=DEV).manual_seed(0)
    q = (torch.randn(B, Nq, Cc, device=DEV, generator=g) * scale).to(torch.bfloat16)
    k32 = torch.randn(B, Nk, Cc, device=DEV, generator=g) * scale
    k = k32.to(torch.bfloat16)
    kpre = (k32 * (1.4426950408889634 / D ** 0.5)).to(torch.bfloat16)     # what the scaled to_k weights produce
    v = torch.randn(B, Nk, Cc, device=DEV, generator=g).to(torch.bfloat16)
    ldvt = (Nk + 7) // 8 * 8
    vt = torch.zeros(B, Cc, ldvt, dtype=torch.bfloat16, device=DEV); vt[:, :, :Nk] = v.permute(0, 2, 1)
    qf, kf, vf = [t.float().view(B, -1, h, D).permute(0, 2, 1, 3) for t in (q, k32, v)]
    ref = torch.nn.functional.scaled_dot_product_attention(qf[:2], kf[:2], vf[:2]).permute(0, 2, 1, 3).reshape(2, Nq, Cc)
    out = []
    for var in variants:
        L.gyre_debug_force_attn_variant(var)
        o = torch.empty(B, Nq, Cc, dtype=torch.bfloat16, device=DEV)
        if var in (3, 5, 6):      # prescaled-K path: 3 = folded v2 kernel, 5 = software-pipelined v3
            L.gyre_debug_force_attn_variant(var)
            run = lambda: L.gyre_op_attention_ex(st(), vp(q), Cc, vp(kpre), Cc, vp(vt), ldvt, B, h, Nq, Nk, D, vp(o), Cc, 1)
        else:
            run = lambda: L.gyre_op_attention(st(), vp(q), Cc, vp(k), Cc, vp(vt), ldvt, B, h, Nq, Nk, D, vp(o), Cc)
        rc = run(); torch.cuda.synchronize()
        err = float((o[:2].float() - ref).norm() / ref.norm())
        us = timeit(run)
        fl = 4.0 * B * h * Nq * Nk * D
        out.append(f"v{var}: rc={rc} err={err:.2e} {us:8.1f} us {fl/us/1e6:6.0f} TF/s")
    print(f"B{B} h{h} Nq{Nq} Nk{Nk} D{D} x{scale}: " + " | ".join(out))
L.gyre_debug_force_attn_variant(0)
