"""Dev tool: do the producer-side GroupNorm statistics permute exactly with the batch?"""
import os, sys, ctypes as C, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gyre_amd import _lib
from gpu_util import DEV, repack_conv, vp, st
L = _lib.lib()
B, HW, Cc, unit = 16, 4096, 320, 10
g = torch.Generator().manual_seed(1)
perm = torch.randperm(B, generator=g)
def run_linear(x, r):
    M = B * HW
    w = (torch.randn(Cc, Cc, generator=torch.Generator().manual_seed(2)) / math.sqrt(Cc)).to(torch.bfloat16).to(DEV)
    b = torch.randn(Cc, generator=torch.Generator().manual_seed(3)).to(DEV)
    y = torch.empty(M, Cc, dtype=torch.bfloat16, device=DEV)
    stats = torch.zeros(M // 16, Cc // unit, 2, device=DEV)
    rows = C.c_int(0)
    _lib.check(L.gyre_op_linear_colstats(st(), vp(x), M, Cc, vp(w), Cc, vp(b), vp(r), HW, unit, vp(y), vp(stats), stats.numel() * 4, None, 0, C.byref(rows)))
    torch.cuda.synchronize()
    n = M // rows.value
    return y.reshape(B, HW, Cc), stats[:n].reshape(B, n // B, Cc // unit, 2).clone(), rows.value
def run_conv(x, r, Cin=320):
    w = repack_conv(torch.randn(Cc, Cin, 3, 3, generator=torch.Generator().manual_seed(4)) / math.sqrt(9 * Cin))
    b = torch.randn(Cc, generator=torch.Generator().manual_seed(5)).to(DEV)
    y = torch.empty(B, 64, 64, Cc, dtype=torch.bfloat16, device=DEV)
    stats = torch.zeros(B * HW // 16, Cc // unit, 2, device=DEV)
    rows = C.c_int(0)
    need = L.gyre_op_gemm_splitk_bytes(1, B * HW, Cc, 9 * Cin, B)
    ws = torch.empty(max(need, 16) + 256, dtype=torch.uint8, device=DEV)
    _lib.check(L.gyre_op_conv3x3_colstats(st(), vp(x), B, 64, 64, Cin, vp(w), Cc, vp(b), vp(r), 1, 0, unit, vp(y), vp(stats), stats.numel() * 4, vp(ws), ws.numel(), C.byref(rows)))
    torch.cuda.synchronize()
    n = B * HW // rows.value
    return y.reshape(B, HW, Cc), stats[:n].reshape(B, n // B, Cc // unit, 2).clone(), rows.value
for name, fn, shape in (("linear", run_linear, (B, HW, Cc)), ("conv", run_conv, (B, 64, 64, Cc)), ("conv_in", lambda x, r: run_conv(x, r, 8), (B, 64, 64, 8))):
    x = torch.randn(*shape, generator=g).to(torch.bfloat16).to(DEV)
    r = torch.randn(B, HW, Cc, generator=g).to(torch.bfloat16).to(DEV)
    y0, s0, rows = fn(x, r)
    y1, s1, _ = fn(x[perm].contiguous(), r[perm].contiguous())
    print(name, "rows", rows, "y permutes:", bool(torch.equal(y1, y0[perm])), "stats permute:", bool(torch.equal(s1, s0[perm])),
          "max stat diff", float((s1 - s0[perm]).abs().max()))
    if not torch.equal(s1, s0[perm]):
        d = (s1 - s0[perm]).abs()
        idx = d.nonzero()[:8]
        print("  first mismatches (sample, chunk, unit, which):", idx.tolist())
