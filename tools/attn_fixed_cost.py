"""Per-workgroup fixed cost of the pipelined attention kernel: time of 16 x 8 heads x 4096 queries x D = 40 against Nk keys
(Nk = 256 ... 4096): slope = per-key-tile time, intercept = what a workgroup pays whatever its key count."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, st, vp
L = _lib.lib()
B, H, Nq, D = 16, 8, 4096, 40
C = H * D
g = torch.Generator(device=DEV).manual_seed(0)
q = torch.randn(B, Nq, C, device=DEV, generator=g).to(torch.bfloat16)
o = torch.empty(B, Nq, C, dtype=torch.bfloat16, device=DEV)
res = []
for rep in range(2):
    for Nk in (4096, 2048, 1024, 512, 256, 4096):
        k = (torch.randn(B, Nk, C, device=DEV, generator=g) * 0.25).to(torch.bfloat16)
        vt = torch.randn(B, C, Nk, device=DEV, generator=g).to(torch.bfloat16)
        L.gyre_debug_force_attn_variant(5)
        run = lambda: _lib.check(L.gyre_op_attention_ex(st(), vp(q), C, vp(k), C, vp(vt), Nk, B, H, Nq, Nk, D, vp(o), C, 1))
        for _ in range(3): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); e.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(e) * 1e3)
        ts.sort()
        print(f"Nk={Nk:5d}: {ts[len(ts) // 2]:7.1f} us  ({Nk // 64} key tiles per workgroup, 4096 workgroups = 8 rounds of 512)", flush=True)
L.gyre_debug_force_attn_variant(0)
