"""Dev tool: exhaustive (tile config x split-K) sweep over every distinct GEMM / conv problem of one SD1.5 UNet forward.

  1. one forward with GYRE_GEMM_DUMP=1 (kernels_gemm.hip) lists the problems with the planner's choice
  2. each distinct plain problem (bf16 row-major output, no fused Q|K|V / transposed output) is re-run standalone through
     gyre_op_linear / gyre_op_conv3x3 with gyre_debug_force_gemm_cfg(cfg | splits << 8) for every valid combination
  3. prints planner time vs best time per problem and the summed headroom over the forward

Usage (GPU box):  [COLD=1|2|3] python tools/gemm_sweep.py [B] [latent] [sd15|sdxl|vae] [extra cfg ids, comma separated]     default 16 64 sd15
(vae: one decode of B latents; sdxl: the SDXL-base topology)
"""
import collections
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if os.environ.get("GYRE_GEMM_DUMP") is None:
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    LAT = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    MODEL = sys.argv[3] if len(sys.argv) > 3 else "sd15"
    EXTRA = tuple(int(c) for c in sys.argv[4].split(",")) if len(sys.argv) > 4 else ()      # e.g. 20,21,22,23: the 4-wave pipelined forms
    env = dict(os.environ, GYRE_GEMM_DUMP="1", SWEEP_B=str(B), SWEEP_LAT=str(LAT), SWEEP_MODEL=MODEL)
    out = subprocess.run([sys.executable, __file__, "dump"], env=env, capture_output=True, text=True)
    shapes = collections.Counter()
    for line in out.stderr.splitlines():
        if line.startswith("GYRE_GEMM "):
            shapes[line[len("GYRE_GEMM "):]] += 1
    if not shapes:
        print(out.stdout[-2000:], out.stderr[-2000:])
        sys.exit(1)
    import torch
    from gyre_amd import _lib
    L = _lib.lib()
    dev = "cuda:0"
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    L.gyre_debug_set_splitk_workspace(C.c_void_p(ws.data_ptr()), ws.numel())
    wblk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)      # blocked weight copies, as the model handles keep them (round 5)
    L.gyre_debug_set_wblk_workspace(C.c_void_p(wblk.data_ptr()), wblk.numel())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(0)

    COLD = os.environ.get("COLD") in ("1", "2", "3")  # 1: evict L2 / Infinity Cache before every timed launch (the in-UNet regime)
    NOFLUSH = os.environ.get("COLD") == "2"      # 2: time launches one by one like 1, but leave the caches alone
    REWARM = os.environ.get("COLD") == "3"       # 3: like 1, then the activation operand is re-written by a copy kernel: what a layer
                                                 #    sees right behind its producer (activations in the Infinity Cache, weights cold)
    rewarm_pair = [None, None]
    if COLD:
        fl_a = torch.empty(384 << 20, dtype=torch.uint8, device=dev)
        fl_b = torch.empty(384 << 20, dtype=torch.uint8, device=dev)

    def timeit(fn, n=12):
        for _ in range(2):
            rc = fn()
            if rc:
                return None
        torch.cuda.synchronize()
        if COLD:
            tot = 0.0
            reps = max(n // 2, 4)
            for _ in range(reps):
                if not NOFLUSH:
                    fl_a.copy_(fl_b)
                    if REWARM and rewarm_pair[0] is not None:
                        rewarm_pair[0].copy_(rewarm_pair[1])
                else:
                    torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                tot += a.elapsed_time(b)
            return tot / reps * 1e3
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3     # us

    tot_plan = tot_best = 0.0
    rows = []
    for desc, cnt in shapes.items():
        f = {k: int(v) for k, v in re.findall(r"(\w+)=(-?\d+)", desc)}
        if f["out"] != 0 or f["vt"] or f["batch"] > 1:
            continue
        M, N, K = f["M"], f["N"], f["K"]
        res = torch.randn(M, (N // 2 if f["geglu"] else N), device=dev, generator=g).to(torch.bfloat16) if f["res"] else None
        dual = (f["mode"] == 0 and f["C1"] != K) or (f["mode"] == 1 and f["C1"] != f["Cin"])
        if f["mode"] == 0:
            x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
            y = torch.empty(M, N // 2 if f["geglu"] else N, dtype=torch.bfloat16, device=dev)
            n_arg = N // 2 if f["geglu"] else N
            call = lambda: L.gyre_op_linear(st, C.c_void_p(x.data_ptr()), M, K, C.c_void_p(w.data_ptr()), n_arg, None,
                                            C.c_void_p(res.data_ptr()) if res is not None else None, f["geglu"], C.c_void_p(y.data_ptr()))
        else:
            Bn = f["samples"] or 1
            x = torch.randn(Bn, f["Hi"], f["Wi"], f["Cin"], device=dev, generator=g).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
            y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            call = lambda: L.gyre_op_conv3x3(st, C.c_void_p(x.data_ptr()), Bn, f["Hi"], f["Wi"], f["Cin"], C.c_void_p(w.data_ptr()), N,
                                             None, C.c_void_p(res.data_ptr()) if res is not None else None, f["stride"], f["ups"], 0,
                                             C.c_void_p(y.data_ptr()))
        if REWARM:
            rewarm_pair[0], rewarm_pair[1] = x, x.clone()
        # dual-source problems (skip concat) are timed as single-source ones of the same shape; the pipelined kernel (24) cannot
        # take them, so it is left out and "planner" is the recorded choice of the real (dual-source) launch
        L.gyre_debug_force_gemm_cfg((f["cfg"] | (f["splits"] << 8)) if dual else 0)
        t_plan = timeit(call)
        best = (t_plan, "planner")
        results = {}
        for cfg in ((1, 2, 3, 4, 5, 6, 7) if dual else (1, 2, 3, 4, 5, 6, 7, 24) + EXTRA):
            for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
                if sp > 1 and (cfg < 4 or f["geglu"] or K // 64 // sp < 4):
                    continue
                if f["geglu"] and cfg in (4, 5, 24):
                    continue
                L.gyre_debug_force_gemm_cfg(cfg | (sp << 8))
                t = timeit(call, 8)
                if t is None:
                    continue
                results[(cfg, sp)] = t
                if t < best[0]:
                    best = (t, f"cfg{cfg} x{sp}")
        L.gyre_debug_force_gemm_cfg(0)
        tot_plan += cnt * t_plan
        tot_best += cnt * best[0]
        top = sorted(results.items(), key=lambda kv: kv[1])[:3]
        rows.append((cnt * (t_plan - best[0]), f"{'conv' if f['mode'] else 'lin '}{'*' if dual else ' '} M={M:6d} N={N:5d} K={K:6d} geglu={f['geglu']} res={f['res']} x{cnt:2d} "
                     f"planner cfg{f['cfg']} x{f['splits']} {t_plan:7.1f} us | best {best[1]:10s} {best[0]:7.1f} us | "
                     + " ".join(f"c{c}x{s}:{t:.0f}" for (c, s), t in top)))
    for gain, line in sorted(rows, key=lambda r: -r[0]):
        print(f"{gain:7.1f} us  {line}")
    print(f"sum over the forward (covered problems): planner {tot_plan / 1e3:.2f} ms, per-problem best {tot_best / 1e3:.2f} ms")
    sys.exit(0)

import torch
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
B, H, dev, model = int(os.environ["SWEEP_B"]), int(os.environ["SWEEP_LAT"]), "cuda:0", os.environ.get("SWEEP_MODEL", "sd15")
if model == "vae":
    vae = GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).load_synthetic(1).to(dev)
    vae.decode(torch.randn(B, 4, H, H, device=dev))
else:
    cfg = gcfg.sdxl_unet() if model == "sdxl" else gcfg.sd15_unet()
    net = GyreHipUNet(cfg).to(torch.bfloat16).load_synthetic(0).to(dev)
    x = torch.randn(B, 4, H, H, device=dev)
    t = torch.full((B,), 500, device=dev)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, device=dev)
    kw = {}
    if model == "sdxl":
        kw = dict(added_cond_kwargs={"text_embeds": torch.randn(B, 1280, device=dev),
                                     "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]], device=dev).expand(B, -1).contiguous()})
    net(x, t, encoder_hidden_states=ctx, **kw)
torch.cuda.synchronize()
