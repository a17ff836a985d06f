"""Round 6, verdict item 6 step A: would Winograd F(2x2, 3x3) be numerically admissible for the stride-1 3x3 convolutions of the UNet's
resnets if the transformed operands are rounded to the 16-bit MFMA input type?  CPU emulation on the fp32 oracle (test infrastructure;
nothing here is product code):

  direct   x, w rounded to bf16 / fp16, fp32 accumulation                        (what the HIP conv kernels do today)
  winograd V = B^T d B from the rounded input, rounded again to the operand type; U = G g G^T from the fp32 master weights, rounded
           once; 16 element-wise GEMMs with fp32 accumulation; Y = A^T M A in fp32  (what a fused Winograd kernel would do)

on the 44 resnet conv1 / conv2 of SD1.5 (every other layer stays fp32, so the figures isolate the convolutions).  Gate A1: whole-UNet
rel-L2 vs the fp32 oracle <= 2e-2 for one CFG evaluation.  Gate A2 (--full): the 51-evaluation config-2 run (committed oracle latents,
tests/golden/full_run_latents.npz) >= 40 dB.  usage: python tools/winograd_gate.py [--full] [--dtype bf16|fp16]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import torch.nn.functional as F
from gyre_amd import config as gcfg, weights
from oracle import models_ref as M, pipeline_ref as PR

HDT = torch.float16 if "fp16" in sys.argv else torch.bfloat16
rnd = lambda t: t.to(HDT).to(torch.float32)
BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
MODE = ["fp32"]
_U = {}


def winograd_conv(x, w, bias):
    B, C, H, W = x.shape
    O = w.shape[0]
    d = F.pad(rnd(x), (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)            # [B, C, H/2, W/2, 4, 4]
    V = rnd(torch.einsum("xa,bchwae,ye->bchwxy", BT, d, BT))                    # B^T d B, rounded to the MFMA operand type
    key = w.data_ptr()
    if key not in _U:
        _U[key] = rnd(torch.einsum("xa,oiae,ye->oixy", G, w, G))                # G g G^T from the fp32 master weights, rounded once
    Mx = torch.einsum("bihwxy,oixy->bohwxy", V, _U[key])                        # 16 GEMMs, fp32 accumulation
    Y = torch.einsum("px,bohwxy,qy->bohpwq", AT, Mx, AT).reshape(B, O, H, W)
    return Y + bias[None, :, None, None]


_orig_conv = M._conv
def patched_conv(x, sd, p, stride=1, padding=1):
    w = sd[p + ".weight"]
    res = (p.endswith(".conv1") or p.endswith(".conv2")) and "resnets" in p and w.shape[-1] == 3 and stride == 1 and padding == 1
    if not res or MODE[0] == "fp32":
        return _orig_conv(x, sd, p, stride, padding)
    if MODE[0] == "direct":
        return F.conv2d(rnd(x), rnd(w), sd.get(p + ".bias"), stride=1, padding=1)
    return winograd_conv(x, w, sd[p + ".bias"])
M._conv = patched_conv

torch.manual_seed(0)
ucfg, vcfg = gcfg.sd15_unet(), gcfg.sd15_vae()
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
g = torch.Generator().manual_seed(7)
x, ctx, t = torch.randn(2, 4, 64, 64, generator=g), torch.randn(2, 77, 768, generator=g), torch.tensor([981, 981])
outs = {}
with torch.no_grad():
    for mode in ("fp32", "direct", "winograd"):
        MODE[0] = mode; t0 = time.time()
        outs[mode] = M.unet_forward(usd, ucfg, x, t, ctx)
        print(f"{mode:9s}: {time.time() - t0:.1f} s", flush=True)
rl = lambda a, b: float((a - b).norm() / b.norm())
print(f"A1 [{str(HDT).split('.')[-1]} operands] whole-UNet rel-L2 vs fp32 oracle (only the 44 resnet 3x3 convs emulated): direct {rl(outs['direct'], outs['fp32']):.3e}   "
      f"winograd F(2x2,3x3) {rl(outs['winograd'], outs['fp32']):.3e}   (gate <= 2e-2)", flush=True)
# per-layer look at one conv: 320 -> 320 at 64x64
w = usd["down_blocks.0.resnets.0.conv1.weight"]; b = usd["down_blocks.0.resnets.0.conv1.bias"]
xi = F.silu(torch.randn(2, 320, 64, 64, generator=g))
ref = F.conv2d(xi, w, b, padding=1)
print(f"   single 320->320 conv on SiLU(N(0,1)) input: direct {rl(F.conv2d(rnd(xi), rnd(w), b, padding=1), ref):.3e}   winograd {rl(winograd_conv(xi, w, b), ref):.3e}", flush=True)

if "--full" in sys.argv:
    import make_full_run_golden as GG
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg))
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_run_latents.npz"))
    run = GG.RUNS["c2"]; text, unc = GG.embeddings(run)
    ref_lat = torch.from_numpy(gold["c2_latents"])
    with torch.no_grad():
        ref_img = (M.vae_decode(vsd, vcfg, ref_lat / 0.18215) / 2 + 0.5).clamp(0, 1)
        for mode in ("winograd", "direct"):
            MODE[0] = mode; t0 = time.time()
            lat, evals = PR.generate_ref(usd, ucfg, vsd, vcfg, text, unc, run["seeds"], 512, 512, run["steps"], 7.5, run["sampler"], decode=False)
            MODE[0] = "fp32"
            img = (M.vae_decode(vsd, vcfg, lat / 0.18215) / 2 + 0.5).clamp(0, 1)
            ps = [PR.psnr(img[i:i + 1], ref_img[i:i + 1]) for i in range(lat.shape[0])]
            print(f"A2 [{str(HDT).split('.')[-1]}] c2 ({evals} evaluations, {time.time() - t0:.0f} s) {mode}: latent rel-L2 {rl(lat, ref_lat):.3e}, image PSNR {['%.1f' % p for p in ps]} dB (gate >= 40)", flush=True)
