import os, sys, subprocess
if os.environ.get("GYRE_PROF_DUMP") is None:
    env = dict(os.environ, GYRE_PROF_DUMP="1")
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    print(out.stdout[-300:], out.stderr[-600:] if out.returncode else "")
    tot, i = 0.0, 0
    for line in out.stderr.splitlines():
        if not line.startswith("GYRE_PROF "): continue
        name, fl, by, us = [x.strip() for x in line[len("GYRE_PROF "):].split("|")]
        t = float(us.split()[0]); tot += t
        print(f"{i:4d} {name:34s} {fl:>14s} {by:>12s} {t:8.1f} us  cum {tot / 1e3:7.3f} ms")
        i += 1
    sys.exit(0)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gyre_amd import config as gcfg, _lib
from gyre_amd.modules import GyreHipVAE
B = int(os.environ.get("B", "8")); dev = "cuda:0"
vae = GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).load_synthetic(1).to(dev)
z = torch.randn(B, 4, 64, 64, device=dev)
for _ in range(2): vae.decode(z)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); vae.decode(z); e1.record(); torch.cuda.synchronize()
print(f"uninstrumented decode B={B}: {e0.elapsed_time(e1):.2f} ms, launches {_lib.lib().gyre_last_launch_count()}")
_lib.prof_enable(None)
vae.decode(z); torch.cuda.synchronize()
_lib.prof_collect(); _lib.prof_enable([])
