"""Round 5: the VAE decoder's 3x3 convs at 512 x 512 (128 -> 128 channels; 256 -> 256 at 256 x 256): time per tile configuration and
with the K loop's loads / MFMAs / epilogue switched off (gyre_debug_gemm_ablation bits 0 - 2; results garbage), warm, B images.
  python tools/vae_conv_probe.py [B]"""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, randn, repack_conv, st, vp
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def wall(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (H, C) in ((512, 128), (256, 256)):
    x = torch.randn(B, H, H, C, device=DEV).to(torch.bfloat16)
    w = repack_conv(randn(C, C, 3, 3, seed=2) / math.sqrt(9 * C))
    b = randn(C, seed=3).to(DEV)
    y = torch.empty(B, H, H, C, dtype=torch.bfloat16, device=DEV)
    fl = 2.0 * B * H * H * 9 * C * C
    by = 2.0 * B * H * H * C * 2
    def run(): _lib.check(L.gyre_op_conv3x3(st(), vp(x), B, H, H, C, vp(w), C, vp(b), None, 1, 0, 0, vp(y)))
    for cfg in (0, 1, 2, 4, 5, 6, 7, 8, 12, 24):
        L.gyre_debug_force_gemm_cfg(cfg)
        try:
            t = wall(run)
            print(f"{H}x{H} C={C} B={B} cfg {cfg:2d}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s  {by / t / 1e6:6.2f} TB/s")
        except Exception as e:
            print(f"{H}x{H} C={C} cfg {cfg}: {str(e)[:80]}")
    L.gyre_debug_force_gemm_cfg(0)
    for bits, name in ((1, "no loads in the K loop"), (2, "no MFMAs"), (4, "no epilogue"), (7, "nothing"), (0x8000000, "two-stage loop (bit 27)")):
        L.gyre_debug_gemm_ablation(bits)
        print(f"{H}x{H} C={C} planner's config, {name}: {wall(run):8.1f} us")
    L.gyre_debug_gemm_ablation(0)
