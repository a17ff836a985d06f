"""Round 5: the nst-deep LDS ring of k_gemm8's linear K loop against the two-stage loop (tuning bit 24) and against the deep ring at one
workgroup per CU (bit 25) on the mid-size linear shapes of the UNet (batch 16): kernel time per launch from the library's HIP events,
warm and (COLD=1) behind a cache-evicting copy (the regime inside the UNet), with a bit-equality check between the loops."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, randn, repack_bias, repack_linear, st, vp
L = _lib.lib()
COLD = os.environ.get("COLD", "1") == "1"
ev_a = torch.empty(300 << 20, dtype=torch.uint8, device=DEV); ev_b = torch.empty(300 << 20, dtype=torch.uint8, device=DEV)
FORCE = [int(c, 0) for c in os.environ.get("FORCE", "0").split(",") if c]
BITS = [int(c, 0) for c in os.environ.get("BITS", "0x1000000,0,0x2000000").split(",") if c]
wsk = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
L.gyre_debug_set_splitk_workspace(vp(wsk), wsk.numel())
if os.environ.get("WBLK", "1") == "1":                  # blocked weight copies for the bare operators (tuning bit 26 = off)
    wblk = torch.empty(128 << 20, dtype=torch.uint8, device=DEV)
    L.gyre_debug_set_wblk_workspace(vp(wblk), wblk.numel())
def timeit(fn, reps=11):
    ts = []
    for _ in range(reps):
        if COLD: ev_b.copy_(ev_a)
        torch.cuda.synchronize()
        _lib.prof_enable(None); fn(); torch.cuda.synchronize()
        c = _lib.prof_collect(); _lib.prof_enable([])
        ts.append(sum(v["ms"] for k_, v in c.items() if "other" not in k_.lower()) * 1e3)      # (the on-the-fly weight blocking is class "other")
    ts.sort(); return ts[len(ts) // 2]
SHAPES = [] if os.environ.get('LINEARS', '1') != '1' else [(4096, 1280, 1280, 1), (4096, 1280, 1280, 0), (16384, 640, 640, 1), (16384, 640, 640, 0), (65536, 320, 320, 1), (65536, 320, 320, 0),
          (16384, 640, 1920, 0), (4096, 1280, 3840, 0), (65536, 1280, 320, 1), (16384, 2560, 640, 1), (4096, 5120, 1280, 1), (1024, 1280, 1280, 1),
          (1024, 5120, 1280, 1), (8192, 640, 640, 1), (2048, 1280, 1280, 1), (32768, 320, 320, 1), (1232, 768, 320, 0), (1232, 768, 1280, 0)]
for (M, K, N, res) in SHAPES:
    x = (randn(M, K, seed=1)).to(torch.bfloat16).to(DEV)
    w, b = repack_linear(randn(N, K, seed=2) / math.sqrt(K)), repack_bias(randn(N, seed=3))
    r = randn(M, N, seed=4).to(torch.bfloat16).to(DEV) if res else None
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    run = lambda: _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), vp(r), 0, vp(y)))
    for force in FORCE:
        out = []; ref = None
        for bits in BITS:
            L.gyre_debug_gemm_ablation(bits); L.gyre_debug_force_gemm_cfg(force)
            try:
                y.zero_(); run(); torch.cuda.synchronize()
                if ref is None: ref = y.clone()
                same = bool(torch.equal(ref, y))
                _lib.prof_enable(None); run(); torch.cuda.synchronize(); names = list(_lib.prof_collect()); _lib.prof_enable([])
                out.append((timeit(run), names, same))
            except Exception as e:
                out.append((float("nan"), [str(e)[:30]], False))
        L.gyre_debug_gemm_ablation(0); L.gyre_debug_force_gemm_cfg(0)
        print(f"M={M:5d} K={K:5d} N={N:5d} res={res} cfg={force:#x}: " + " | ".join(f"{t:6.1f} us {n[0][:24] if n else ''}{'' if s else ' BITS DIFFER'}" for t, n, s in out), flush=True)

# 3x3 convs of the UNet (batch 16) and of the VAE decoder
from gpu_util import repack_conv
CB = int(os.environ.get("CONV_BATCH", "16"))          # batch of the UNet convs below (8: the reverse sweep of config 5)
CONVS = [(16, 64, 320, 320), (16, 64, 640, 320), (16, 32, 640, 640), (16, 32, 1280, 640), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 8, 1280, 1280),
         (16, 32, 320, 640), (16, 16, 640, 1280), (16, 32, 1920, 640), (16, 64, 960, 320), (4, 128, 512, 512), (2, 256, 256, 256)]
if os.environ.get("CONVS", "1") == "1":
    for (B, H, Ci, Co) in [((CB if B_ == 16 else B_), H_, Ci_, Co_) for (B_, H_, Ci_, Co_) in CONVS]:
        x = randn(B, H, H, Ci, seed=5).to(torch.bfloat16).to(DEV)
        w = repack_conv(randn(Co, Ci, 3, 3, seed=6) / math.sqrt(9 * Ci)); b = repack_bias(randn(Co, seed=7))
        y = torch.empty(B, H, H, Co, dtype=torch.bfloat16, device=DEV)
        run = lambda: _lib.check(L.gyre_op_conv3x3(st(), vp(x), B, H, H, Ci, vp(w), Co, vp(b), None, 1, 0, 0, vp(y)))
        out = []; ref = None
        for bits, force in [(b_, f_) for f_ in FORCE for b_ in BITS]:
            L.gyre_debug_gemm_ablation(bits); L.gyre_debug_force_gemm_cfg(force)
            try:
                y.zero_(); run(); torch.cuda.synchronize()
                if ref is None: ref = y.clone()
                same = bool(torch.equal(ref, y))
                _lib.prof_enable(None); run(); torch.cuda.synchronize(); names = [n for n in _lib.prof_collect() if "other" not in n.lower()]; _lib.prof_enable([])
                out.append((timeit(run), names, same))
            except Exception as e:
                out.append((float("nan"), [str(e)[:30]], False))
        L.gyre_debug_gemm_ablation(0); L.gyre_debug_force_gemm_cfg(0)
        fl = 2.0 * B * H * H * Co * 9 * Ci
        print(f"conv B={B} {H}x{H} {Ci}->{Co}: " + " | ".join(f"{t:6.1f} us {fl / t / 1e6:5.0f} TF {n[0][:22] if n else ''}{'' if s_ else ' BITS DIFFER'}" for t, n, s_ in out), flush=True)
