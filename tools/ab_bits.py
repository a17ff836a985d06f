"""Dev tool: same-process interleaved A/B of one gyre_debug_gemm_ablation bit mask on the UNet forward.
  python tools/ab_bits.py <mask hex> [B ...]      e.g.  python tools/ab_bits.py 0x10000000 16 8 4 2
Prints ms per forward with the mask clear / set (alternating blocks of 5 forwards, 6 rounds) and whether the outputs are equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg, _lib
from gyre_amd.modules import GyreHipUNet, cfg_pairs
mask = int(sys.argv[1], 16)
Bs = [int(a) for a in sys.argv[2:]] or [16, 2]
dev = "cuda:0"
L = _lib.lib()
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in net.named_parameters():
        if p.ndim > 1: p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
        elif k.endswith("weight"): p.fill_(1.0)
        else: p.zero_()
net._invalidate()
for B in Bs:
    x = torch.randn(B // 2, 4, 64, 64, device=dev, generator=g); x = torch.cat([x, x]); ctx = torch.randn(B, 77, 768, device=dev, generator=g)
    def run():
        with cfg_pairs():
            return net(x, 500, encoder_hidden_states=ctx).sample
    outs = {}
    for m in (0, mask):
        L.gyre_debug_gemm_ablation(m)
        for _ in range(3): outs[m] = run()
    torch.cuda.synchronize()
    t = {0: [], mask: []}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(6):
        for m in (0, mask):
            L.gyre_debug_gemm_ablation(m)
            run(); torch.cuda.synchronize()
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            t[m].append(e0.elapsed_time(e1) / 5)
    L.gyre_debug_gemm_ablation(0)
    med = {m: sorted(v)[len(v) // 2] for m, v in t.items()}
    print(f"B={B}: mask clear {med[0]:.3f} ms, mask {mask:#x} set {med[mask]:.3f} ms (min {min(t[0]):.3f} / {min(t[mask]):.3f}); "
          f"outputs equal: {bool(torch.equal(outs[0], outs[mask]))}")
