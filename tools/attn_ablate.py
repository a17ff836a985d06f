"""Where the pipelined attention kernel's time goes (64x64 self-attention of the SD1.5 UNet at batch 16: 16 x 8 heads x 4096^2 x 40):
timing ablations compiled in with GYRE_ATTN_ABLATIONS (touch gyre_amd/csrc/kernels_attn.hip && GYRE_ATTN_ABLATIONS=1 python -c
'from gyre_amd import build as b; b.build()').  Results of the ablated runs are garbage, their times are what matters."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, st, vp
L = _lib.lib()
B, H, N, D = int(os.environ.get("B", 16)), 8, int(os.environ.get("N", 4096)), 40
C = H * D
g = torch.Generator(device=DEV).manual_seed(0)
q = torch.randn(B, N, C, device=DEV, generator=g).to(torch.bfloat16)
k = (torch.randn(B, N, C, device=DEV, generator=g) * 0.25).to(torch.bfloat16)
vt = torch.randn(B, C, N, device=DEV, generator=g).to(torch.bfloat16)
o = torch.empty(B, N, C, dtype=torch.bfloat16, device=DEV)
run = lambda: _lib.check(L.gyre_op_attention_ex(st(), vp(q), C, vp(k), C, vp(vt), N, B, H, N, N, D, vp(o), C, 1))
names = {0: "full", 1: "no exponentials", 2: "no tile requests in the loop", 4: "no barrier / wait", 6: "no requests, no barrier", 8: "no MFMA",
         16: "no fragment reads", 32: "no bf16 packing", 33: "no exp, no packing", 9: "no exp, no MFMA", 24: "no MFMA, no fragment reads",
         57: "no exp / pack / MFMA / reads", 63: "nothing but the loop skeleton", 22: "no requests / barrier / fragment reads", 7 << 0: "always checked (variant 7)"}
for abl in (0, 7, 1, 32, 33, 8, 16, 24, 9, 2, 4, 6, 22, 57, 63, 0):
    L.gyre_debug_force_attn_variant(7 if abl == 7 else abl << 8)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    ts.sort()
    print(f"abl {abl:2d} {names.get(abl, ''):42s} {ts[len(ts) // 2]:7.1f} us (min {ts[0]:.1f})", flush=True)
L.gyre_debug_force_attn_variant(0)
