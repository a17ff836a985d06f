"""Dev tool: interleaved A/B timing of the native SD1.5 UNet (and VAE decode) forward under gyre_debug_gemm_ablation
flag values, e.g. `python tools/ab_unet.py 0 0x400` (0x400 = planner without the pipelined 32x32x16 configs).
Reports the median over rounds per variant (guide rule 24: within-process interleaved rounds)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg, _lib
from gyre_amd.modules import GyreHipUNet, GyreHipVAE

flags = [int(a, 0) for a in sys.argv[1:]] or [0, 0x400]
B = int(os.environ.get("B", "16"))
H = int(os.environ.get("LAT", "64"))
rounds = int(os.environ.get("ROUNDS", "5"))
dev = "cuda:0"
L = _lib.lib()


def fill(m):
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.ndim > 1:
                p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
            elif k.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    m._invalidate()


net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev); fill(net)
vae = GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).to(dev); fill(vae)
x = torch.randn(B, 4, H, H, device=dev); t = torch.full((B,), 500, device=dev); ctx = torch.randn(B, 77, 768, device=dev)
z = torch.randn(max(1, B // 2), 4, H, H, device=dev)


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


res = {f: ([], []) for f in flags}
outs = {}
for r in range(rounds):
    for f in flags:
        L.gyre_debug_gemm_ablation(f & 0xfffffff)
        L.gyre_debug_force_attn_variant((f >> 28) & 15)      # bits 28-31: attention variant (7 = per-tile check in every tile)
        res[f][0].append(timeit(lambda: net(x, t, encoder_hidden_states=ctx).sample, 5))
        res[f][1].append(timeit(lambda: vae.decode(z).sample, 2))
        if r == 0:
            outs[f] = (net(x, t, encoder_hidden_states=ctx).sample.clone(), L.gyre_last_launch_count())
L.gyre_debug_gemm_ablation(0)
L.gyre_debug_force_attn_variant(0)
for f in flags:
    u, v = res[f]
    same = bool(torch.equal(outs[f][0], outs[flags[0]][0]))
    print(f"flags {f:#x}: UNet B={B} lat {H}: median {statistics.median(u):.2f} ms (min {min(u):.2f}) = {B * 0.803 * (H / 64) ** 2 / statistics.median(u):.0f} TFLOP/s eff | "
          f"VAE decode B={z.shape[0]}: {statistics.median(v):.2f} ms | launches {outs[f][1]} | bits equal to first variant: {same}")
