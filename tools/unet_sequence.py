"""Dev tool: the launches of ONE SD1.5 UNet forward in issue order with their HIP-event times (GYRE_PROF_DUMP lines),
so that time can be attributed to levels / blocks.  B=16 LAT=64 python tools/unet_sequence.py > out.txt"""
import os, sys, subprocess
if os.environ.get("GYRE_PROF_DUMP") is None:
    env = dict(os.environ, GYRE_PROF_DUMP="1")
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    print(out.stdout[-300:], out.stderr[-600:] if out.returncode else "")
    tot, i = 0.0, 0
    for line in out.stderr.splitlines():
        if not line.startswith("GYRE_PROF "):
            continue
        name, fl, by, us = [x.strip() for x in line[len("GYRE_PROF "):].split("|")]
        t = float(us.split()[0]); tot += t
        print(f"{i:4d} {name:34s} {fl:>14s} {by:>12s} {t:8.1f} us  cum {tot / 1e3:7.3f} ms")
        i += 1
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg, _lib
from gyre_amd.modules import GyreHipUNet
B = int(os.environ.get("B", "16")); H = int(os.environ.get("LAT", "64")); dev = "cuda:0"; L = _lib.lib()
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in net.named_parameters():
        if p.ndim > 1: p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
        elif k.endswith("weight"): p.fill_(1.0)
        else: p.zero_()
net._invalidate()
L.gyre_debug_gemm_ablation(int(os.environ.get("FLAGS", "0"), 0))          # planner debug bits (include/gyre_hip.h)
x = torch.randn(B, 4, H, H, device=dev); t = 500; ctx = torch.randn(B, 77, 768, device=dev)
PAIRS = os.environ.get("PAIRS") == "1"               # CFG-parallel batch: halves share their prefix (modules.cfg_pairs)
if PAIRS:
    x = torch.cat([x[: B // 2], x[: B // 2]])
from gyre_amd.modules import cfg_pairs
import contextlib
def run():
    with (cfg_pairs() if PAIRS else contextlib.nullcontext()):
        return net(x, t, encoder_hidden_states=ctx)
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f"uninstrumented call B={B}: {e0.elapsed_time(e1):.2f} ms, launches {L.gyre_last_launch_count()}")
_lib.prof_enable(None)
run(); torch.cuda.synchronize()
_lib.prof_collect()
