"""Round 5: the 3x3 convs of the 8x8 / 16x16 UNet levels (1280 -> 1280, 2560 -> 1280) at batch B: time per (tile config, K slices, ring
switch), cold-ish (a 512 MB copy between repetitions evicts the Infinity Cache), incl. the split-K reduction.
  python tools/deep_conv_probe.py [B]"""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, randn, repack_conv, st, vp
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ws = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
L.gyre_debug_set_splitk_workspace(vp(ws), ws.numel())
junk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV); junk2 = torch.empty_like(junk)
def timed(fn, n=7):
    ts = []
    for _ in range(n):
        junk2.copy_(junk); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for (H, Cin, Cout) in ((8, 1280, 1280), (8, 2560, 1280), (16, 1280, 1280), (16, 2560, 1280)):
    x = torch.randn(B, H, H, Cin, device=DEV).to(torch.bfloat16)
    w = repack_conv(randn(Cout, Cin, 3, 3, seed=2) / math.sqrt(9 * Cin))
    b = randn(Cout, seed=3).to(DEV)
    y = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device=DEV)
    def run(): _lib.check(L.gyre_op_conv3x3(st(), vp(x), B, H, H, Cin, vp(w), Cout, vp(b), None, 1, 0, 0, vp(y)))
    L.gyre_debug_force_gemm_cfg(0); L.gyre_debug_gemm_ablation(0)
    print(f"{H}x{H} {Cin}->{Cout} B={B}: planner {timed(run):7.1f} us")
    for cfg, sp, bits in ((8, 8, 0), (8, 4, 0), (8, 4, 0x2000000), (8, 6, 0), (8, 2, 0), (8, 16, 0), (24, 4, 0), (24, 8, 0), (24, 11, 0), (24, 16, 0), (4, 8, 0), (5, 8, 0)):
        L.gyre_debug_force_gemm_cfg(cfg | (sp << 8)); L.gyre_debug_gemm_ablation(bits)
        try:
            print(f"    cfg {cfg:2d} x{sp:2d} slices bits {bits:#x}: {timed(run):7.1f} us")
        except Exception as e:
            print(f"    cfg {cfg} x{sp}: {str(e)[:70]}")
L.gyre_debug_force_gemm_cfg(0); L.gyre_debug_gemm_ablation(0); L.gyre_debug_set_splitk_workspace(None, 0)
