"""Dev tool: per-launch (per layer shape) HIP-event times of one SD1.5 UNet forward, grouped by (class, flops, bytes).
GYRE_PROF_DUMP=1 python tools/unet_layers.py 2>&1 | ... (the script sets the variable itself)."""
import os, sys, subprocess, collections
if os.environ.get("GYRE_PROF_DUMP") is None:
    env = dict(os.environ, GYRE_PROF_DUMP="1")
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    rows = collections.OrderedDict()
    for line in out.stderr.splitlines():
        if not line.startswith("GYRE_PROF "): continue
        name, fl, by, us = [x.strip() for x in line[len("GYRE_PROF "):].split("|")]
        key = (name, fl, by)
        rows.setdefault(key, []).append(float(us.split()[0]))
    tot = sum(sum(v) for v in rows.values())
    print(out.stdout[-400:], out.stderr[-600:] if out.returncode else '')
    print(f"total timed {tot / 1e3:.2f} ms")
    # one row per distinct (kernel class, algorithmic FLOPs, algorithmic bytes) = per layer shape: fractions of the 2.5 PFLOP/s dense
    # bf16 MFMA peak and of the 8 TB/s HBM peak, and the time the shape spends ABOVE its own roof ("gap": what tuning could win at
    # most) - sorted by that gap.  MD=path writes the table as markdown (profiles/rNN_unet_layers.md).
    table = []
    for (name, fl, by), v in rows.items():
        mf = float(fl.split()[0]); kb = float(by.split()[0]); t = sum(v) / len(v)
        tf, tbs = mf / t, kb / t / 1e3
        roof = max(tf / 2500.0, tbs / 8.0)
        table.append((sum(v) * (1.0 - min(roof, 1.0)), name, len(v), sum(v), t, tf, tbs, roof, fl, by))
    table.sort(reverse=True)
    for gap, name, n, st, t, tf, tbs, roof, fl, by in table:
        print(f"{name:30s} x{n:3d}  {st:8.1f} us tot  {t:7.1f} us each  {tf:7.0f} TF ({tf / 2500:.3f})  {tbs:6.2f} TB/s ({tbs / 8:.3f})  "
              f"gap {gap:7.1f} us  [{fl}, {by}]")
    md = os.environ.get("MD")
    if md:
        with open(md, "w") as f:
            f.write(f"Per-shape roofline of one UNet forward (`tools/unet_layers.py`, HIP events around every launch; B = "
                    f"{os.environ.get('B', '16')}, latent {os.environ.get('LAT', '64')}; sum of timed launches {tot / 1e3:.2f} ms).  "
                    "`mfma` = algorithmic FLOP/s over 2.5 PFLOP/s, `hbm` = algorithmic bytes/s over 8 TB/s, `gap` = time above the "
                    "shape's own roof (total time x (1 - max of the two fractions)); sorted by gap.\n\n")
            f.write("| kernel class | launches | us total | us each | GFLOP | MB | TFLOP/s | mfma | TB/s | hbm | gap us |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
            for gap, name, n, st, t, tf, tbs, roof, fl, by in table:
                f.write(f"| `{name}` | {n} | {st:.1f} | {t:.1f} | {float(fl.split()[0]) / 1e3:.1f} | {float(by.split()[0]) / 1e3:.1f} | {tf:.0f} | "
                        f"{tf / 2500:.3f} | {tbs:.2f} | {tbs / 8:.3f} | {gap:.1f} |\n")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg, _lib
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
B = int(os.environ.get("B", "16")); H = int(os.environ.get("LAT", "64")); dev = "cuda:0"; L = _lib.lib()
VAE = os.environ.get("MODEL") == "vae"                      # MODEL=vae B=8: one decode of B latents instead of the UNet
net = (GyreHipVAE(gcfg.sd15_vae()) if VAE else GyreHipUNet(gcfg.sd15_unet())).to(torch.bfloat16).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in net.named_parameters():
        if p.ndim > 1: p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
        elif k.endswith("weight"): p.fill_(1.0)
        else: p.zero_()
net._invalidate()
x = torch.randn(B, 4, H, H, device=dev); t = torch.full((B,), 500, device=dev); ctx = torch.randn(B, 77, 768, device=dev)
L.gyre_debug_gemm_ablation(int(sys.argv[1], 0) if len(sys.argv) > 1 else 0)
run = (lambda: net.decode(x)) if VAE else (lambda: net(x, t, encoder_hidden_states=ctx))
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f"uninstrumented call: {e0.elapsed_time(e1):.2f} ms")
_lib.prof_enable(None)
run(); torch.cuda.synchronize()
_lib.prof_collect()
print("done")
