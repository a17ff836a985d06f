"""Dev probe: the optimistic attention pass and its checked redo, one configuration per subprocess (a GPU fault aborts the process)."""
import math, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) == 1:
    for D in (40, 64, 32, 80, 16):
        for excess in (0.0, 60.0, 90.0, 300.0):
            for variant in (0, 7):
                for N in (640, 1024):
                    r = subprocess.run([sys.executable, __file__, str(D), str(excess), str(variant), str(N)], capture_output=True, text=True)
                    print(f"D={D} excess={excess} variant={variant} N={N}: rc={r.returncode} {r.stdout.strip()[-200:]} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
    sys.exit(0)
import torch
from gyre_amd import _lib
from gpu_util import DEV, bf16_round, randn, rel_l2, st, vp
D, excess, variant, N = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
L = _lib.lib()
B, heads = 1, 2
C_ = heads * D
c = 1.4426950408889634 / math.sqrt(D)
q = bf16_round(randn(B, N, C_, seed=173))
k32 = randn(B, N, C_, seed=174)
if excess:
    qn = q[0, 5, :D]
    k32[0, N - 300, :D] = qn * (excess / c / float(qn @ qn))
v = bf16_round(randn(B, N, C_, seed=175))
sp = lambda t: t.reshape(B, t.shape[1], heads, D).permute(0, 2, 1, 3)
ref = ((sp(q) @ sp(k32).transpose(-1, -2)) * D ** -0.5).softmax(-1) @ sp(v)
ref = ref.permute(0, 2, 1, 3).reshape(B, N, C_)
kpre = (k32 * c).to(torch.bfloat16).to(DEV)
vt = v.permute(0, 2, 1).to(torch.bfloat16).contiguous().to(DEV)
o = torch.full((B, N, C_), float("nan"), dtype=torch.bfloat16, device=DEV)
L.gyre_debug_force_attn_variant(variant)
_lib.check(L.gyre_op_attention_ex(st(), vp(q.to(torch.bfloat16).to(DEV)), C_, vp(kpre), C_, vp(vt), N, B, heads, N, N, D, vp(o), C_, 1))
torch.cuda.synchronize()
print(f"finite={bool(torch.isfinite(o).all())} rel_l2={rel_l2(o.float().cpu(), ref):.3e}")
