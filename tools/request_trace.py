"""Dev tool: what a ONE-image request spends outside the UNet / VAE kernels.  Under rocprofv3:
  cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/rq -o rq -- python tools/request_trace.py run
  python tools/request_trace.py parse /tmp/rq
parse: per kernel name (library kernels `k_*` folded into one line) launches and GPU time of the LAST request, idle time
between launches, and the wall time from the first to the last launch."""
import csv, glob, os, re, sys
from collections import defaultdict
if sys.argv[1] == "parse":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
    marks = [i for i, r in enumerate(rows) if "FillFunctor<double>" in r[2]]
    seg = rows[marks[-2] + 1:marks[-1]]
    wall = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    acc = defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        n = re.sub(r"^void ", "", n)
        key = "gyre k_* kernels" if n.startswith("k_") else n.split("(")[0][:110]
        acc[key][0] += 1; acc[key][1] += e - s
    print(f"{len(seg)} launches, first-to-last {wall / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms, idle {(wall - busy) / 1e6:.2f} ms")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{n:6d} {t / 1e3:10.1f} us  {k}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import fill_synthetic_on_device
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gyre_amd.pipeline import GyrePipeline
from gyre_amd.text import ClipTextEncoder, empty_prompt_ids, synthetic_prompt_ids
dev = torch.device("cuda:0")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
unet = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev); fill_synthetic_on_device(unet, 0)
vae = GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).to(dev); fill_synthetic_on_device(vae, 1)
clip = ClipTextEncoder.synthetic(dev, torch.bfloat16, seed=2)
pipe = GyrePipeline(unet, vae, clip, device=dev)
ids, neg = synthetic_prompt_ids(B, seed=1234).to(dev), empty_prompt_ids(B).to(dev)
def request():
    lat = pipe(seeds=[420420420 + j for j in range(B)], height=512, width=512, num_inference_steps=50, guidance_scale=7.5, sampler="dpmpp_2m",
               output_type="latent", input_ids=ids, negative_ids=neg, generator_device="cpu")
    return pipe.vae_decode(lat)
request(); torch.cuda.synchronize()
m = torch.zeros(1, dtype=torch.float64, device=dev)
m.fill_(1.0)
request()
m.fill_(2.0)
torch.cuda.synchronize()
