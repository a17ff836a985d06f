"""Dev tool: how much of a UNet forward the GPU idles between dependent launches.
Run under rocprofv3 (kernel trace), then point the script at the trace:
  cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/q -o q -- python tools/unet_gap.py run [B]
  python tools/unet_gap.py parse /tmp/q [evals]
"""
import csv, glob, os, sys
if sys.argv[1] == "parse":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    evals = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
    # the timed region is bracketed by two marker launches of the elementwise fill kernel on a 1-element tensor
    marks = [i for i, r in enumerate(rows) if "k_gap_marker" in r[2] or "FillFunctor<double>" in r[2]]
    a, b = marks[-2], marks[-1]
    seg = rows[a + 1:b]
    busy = sum(e - s for s, e, _ in seg)
    wall = seg[-1][1] - seg[0][0]
    gaps = sorted((seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1))
    print(f"{len(seg)} launches over {evals} forwards = {len(seg) / evals:.0f} per forward; wall {wall / evals / 1e6:.3f} ms, kernels busy "
          f"{busy / evals / 1e6:.3f} ms per forward, idle {100 * (1 - busy / wall):.1f} %; gap median {gaps[len(gaps) // 2] / 1e3:.2f} us, "
          f"p90 {gaps[int(len(gaps) * .9)] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet, cfg_pairs
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = "cuda:0"
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in net.named_parameters():
        if p.ndim > 1: p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
        elif k.endswith("weight"): p.fill_(1.0)
        else: p.zero_()
net._invalidate()
x = torch.randn(B // 2, 4, 64, 64, device=dev); x = torch.cat([x, x]); ctx = torch.randn(B, 77, 768, device=dev)
def run():
    with cfg_pairs():
        return net(x, 500, encoder_hidden_states=ctx).sample
for _ in range(3): run()
torch.cuda.synchronize()
m = torch.zeros(1, dtype=torch.float64, device=dev)
m.fill_(1.0)
for _ in range(20): run()
m.fill_(2.0)
torch.cuda.synchronize()
