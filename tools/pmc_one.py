"""Run one conv / linear shape a few times (for rocprofv3 --pmc passes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
kind = sys.argv[1]
if os.environ.get("FORCE_CFG"): L.gyre_debug_force_gemm_cfg(int(os.environ["FORCE_CFG"], 0))      # tile config under test
if os.environ.get("GEMM_ABL"): L.gyre_debug_gemm_ablation(int(os.environ["GEMM_ABL"], 0))
if kind == "conv":
    B, H, W, Ci, Co = map(int, sys.argv[2:7])
    x, w, b = rnd(B, H, W, Ci), rnd(Co, 9 * Ci), torch.zeros(Co, device=DEV)
    y = torch.empty(B, H, W, Co, dtype=torch.bfloat16, device=DEV)
    for _ in range(5):
        L.gyre_op_conv3x3(st(), vp(x), B, H, W, Ci, vp(w), Co, vp(b), None, 1, 0, 0, vp(y))
elif kind == "linear":
    M, K, N = map(int, sys.argv[2:5])
    x, w, b = rnd(M, K), rnd(N, K), torch.zeros(N, device=DEV)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    for _ in range(5):
        L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), None, 0, vp(y))
elif kind == "geglu":          # M K F [AR=1: A-resident kernel]
    M, K, F_ = map(int, sys.argv[2:5])
    x, w, b = rnd(M, K), rnd(2 * F_, K), torch.zeros(2 * F_, device=DEV)
    y = torch.empty(M, F_, dtype=torch.bfloat16, device=DEV)
    arws = torch.empty(2 * F_ * K * 2, dtype=torch.uint8, device=DEV)
    if os.environ.get("AR"): L.gyre_debug_set_ar_workspace(vp(arws), arws.numel())
    for _ in range(5):
        L.gyre_op_linear(st(), vp(x), M, K, vp(w), F_, vp(b), None, 1, vp(y))
elif kind == "attn":
    B, h, Nq, Nk, D = map(int, sys.argv[2:7])
    Cc = h * D
    q, k, vt = rnd(B, Nq, Cc), rnd(B, Nk, Cc), rnd(B, Cc, Nk)
    o = torch.empty(B, Nq, Cc, dtype=torch.bfloat16, device=DEV)
    pre = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    for _ in range(5):
        L.gyre_op_attention_ex(st(), vp(q), Cc, vp(k), Cc, vp(vt), Nk, B, h, Nq, Nk, D, vp(o), Cc, pre)
torch.cuda.synchronize()
