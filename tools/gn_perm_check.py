"""Dev tool: does the GroupNorm that takes producer statistics permute exactly with the batch?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gyre_amd import _lib
from gpu_util import DEV, vp, st
L = _lib.lib()
B, H, W, unit = 16, 64, 64, 10
HW = H * W
def cs_ref(t, rows):
    C = t.shape[-1]
    v = t.double().reshape(-1, rows, C // unit, unit)
    return torch.stack([v.sum(dim=(1, 3)), (v * v).sum(dim=(1, 3))], dim=-1).float().contiguous()
g = torch.Generator().manual_seed(3)
perm = torch.randperm(B, generator=g)
for C1, C2 in ((640, 320), (320, 320), (320, 0)):
    C = C1 + C2
    xa = (torch.randn(B, HW, C1, generator=g) * 1.3 + 0.2).to(torch.bfloat16)
    xb = (torch.randn(B, HW, C2, generator=g) * 0.7).to(torch.bfloat16) if C2 else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    def run(a, b):
        ca = cs_ref(a.float().reshape(-1, C1), 256).to(DEV)
        cb = cs_ref(b.float().reshape(-1, C2), 256).to(DEV) if b is not None else None
        ws = torch.empty(L.gyre_op_groupnorm_workspace(B, HW, C, 32) + 256, dtype=torch.uint8, device=DEV)
        y = torch.empty(B, HW, C, dtype=torch.bfloat16, device=DEV)
        ad, bd = a.to(DEV), (b.to(DEV) if b is not None else None)
        _lib.check(L.gyre_op_groupnorm_colstats(st(), vp(ad), vp(bd), C1, B, HW, C, 32, vp(gamma.to(DEV)), vp(beta.to(DEV)), 1e-5, 1,
                                                vp(ca), HW // 256, vp(cb), HW // 256 if b is not None else 0, unit, vp(ws), ws.numel(), vp(y)))
        torch.cuda.synchronize()
        return y
    y0 = run(xa, xb)
    y1 = run(xa[perm].contiguous(), xb[perm].contiguous() if xb is not None else None)
    print(C1, C2, "permutes exactly:", bool(torch.equal(y1, y0[perm])))
