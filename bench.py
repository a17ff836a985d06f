#!/usr/bin/env python
"""Headline benchmark: SD1.5 512x512, 50-step DPM++2M (51 UNet evals, CFG 7.5) images/sec on
N MI355X of one node (BASELINE.json metric; config 2: batch 8 per GPU, bf16 compute).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the whole hot path over one batch of synthetic input: CLIP text encode ->
51 CFG UNet evaluations (batch 16) driven by the DPM-Solver++(2M) sampler -> VAE decode of the 8
images, all inputs resident in HBM.  Each rank generates its own 8 images (weak scaling); the only
collective is the all_gather of the finished latents over RCCL.  Rank 0 prints ONE JSON line.

roofline: the MFMA kernel classes (8-wave GEMM/conv tiles, attention) are timed live with HIP events
on the launch stream during the timed region (gyre_prof_* in the C ABI); the class with the largest
total is reported: achieved = algorithmic FLOPs of its launches / their summed duration, against the
2.5 PFLOP/s dense bf16 MFMA peak; traffic = PMC-derived HBM bytes per launch from profiles/traffic.json.  cpu_baseline: the fp32 oracle (same ATen CPU ops as the reference's CPU path) timed on
the host cores for one CFG UNet evaluation + one VAE decode and extrapolated to a 51-eval image.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

UNET_TFLOP_PER_SAMPLE = 0.803   # 401.6 GMAC @ 64x64 latents
VAE_DEC_TFLOP = 2.515           # 1257 GMAC @ 512^2
MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16
# kernel classes timed live during the timed region; the one with the largest total is reported as the dominant kernel
CANDIDATES = ["k_gemm8<", "k_gemm4s<", "k_attn"]


def fill_synthetic_on_device(module, seed):
    """Random-init weights of the exact architecture, drawn on the GPU (values are irrelevant to
    throughput; fan-in scaling keeps activations finite)."""
    g = torch.Generator(device=next(module.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for k, p in module.named_parameters():
            if p.ndim > 1:
                fan = p[0].numel()
                p.copy_(torch.randn(p.shape, device=p.device, generator=g, dtype=torch.float32) / fan ** 0.5)
            elif "norm" in k and k.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    module._invalidate()


def cpu_baseline(threads):
    """Bounded CPU sample of the same workload with the oracle (kind = "port")."""
    from gyre_amd import config as gcfg, weights
    from oracle import models_ref as M
    ucfg, vcfg = gcfg.sd15_unet(), gcfg.sd15_vae()
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg, encoder=False))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 64, 64, generator=g)
    ctx = torch.randn(2, 77, 768, generator=g)
    t = torch.tensor([981, 981])
    with torch.no_grad():
        t0 = time.perf_counter()
        M.unet_forward(usd, ucfg, x, t, ctx)
        t_unet = time.perf_counter() - t0
        t0 = time.perf_counter()
        M.vae_decode(vsd, vcfg, x[:1])
        t_vae = time.perf_counter() - t0
    per_image = 51 * t_unet + t_vae
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"fp32 oracle: 1 CFG UNet eval (batch 2, 64x64 latents) = {t_unet:.2f} s and 1 VAE decode "
                      f"(512x512) = {t_vae:.2f} s on {threads} threads, extrapolated to 51 evals + 1 decode per image "
                      f"({per_image:.0f} s/image)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class (adds event overhead)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gyre_amd import _lib, config as gcfg
    from gyre_amd.modules import GyreHipUNet, GyreHipVAE
    from gyre_amd.pipeline import GyrePipeline
    from gyre_amd.sharding import gather_batches
    from gyre_amd.text import ClipTextEncoder, empty_prompt_ids, synthetic_prompt_ids

    unet = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
    fill_synthetic_on_device(unet, 0)
    vae = GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).to(dev)
    fill_synthetic_on_device(vae, 1)
    clip = ClipTextEncoder.synthetic(dev, torch.bfloat16, seed=2)
    pipe = GyrePipeline(unet, vae, clip, device=dev)

    B = args.batch
    ids = synthetic_prompt_ids(B, seed=1234 + rank).to(dev)
    neg = empty_prompt_ids(B).to(dev)
    sizes = [B] * world

    def step(i):
        seeds = [420420420 + rank * 100000 + i * B + j for j in range(B)]
        latents = pipe(seeds=seeds, input_ids=ids, negative_ids=neg, height=args.size, width=args.size,
                       num_inference_steps=args.inference_steps, guidance_scale=7.5, sampler="dpmpp_2m",
                       output_type="latent")
        images = pipe.vae_decode(latents)
        if world > 1:
            latents = gather_batches(latents, sizes)  # RCCL all_gather of the finished latents
        return images, latents

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(-1 - i)
    barrier()
    _lib.prof_enable(None if args.profile_all else CANDIDATES)
    step_times = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        s0 = time.perf_counter()
        images, latents = step(i)
        torch.cuda.synchronize()
        step_times.append(time.perf_counter() - s0)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = _lib.prof_collect()
    _lib.prof_enable([])
    assert bool(torch.isfinite(images).all()), "non-finite output"
    evals = pipe.last_unet_evals
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        total_images = world * B * args.steps
        value = total_images / elapsed
        cands = {k: v for k, v in prof.items() if any(k.startswith(c) for c in CANDIDATES)}
        dom_name = max(cands, key=lambda k: cands[k]["ms"]) if cands else ""
        d = cands.get(dom_name)
        roof = None
        if d and d["ms"] > 0:
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            traffic = None
            tj = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tj):
                hits = [v for k, v in json.load(open(tj)).get("kernels", {}).items() if k.startswith(dom_name)]
                n = sum(h["launches_in_pmc_run"] for h in hits)
                if n:
                    traffic = round(sum(h["hbm_bytes_per_launch"] * h["launches_in_pmc_run"] for h in hits) / n)
            roof = {"bound": "mfma", "kernel": dom_name + ", ...>",
                    "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "launches": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
                    "flops_per_launch": d["flops"] / d["launches"],
                    "share_of_step_time": round(d["ms"] * 1e-3 / (elapsed / 1.0), 4)}
        alg_tflop_per_step = B * (evals * 2 * UNET_TFLOP_PER_SAMPLE + VAE_DEC_TFLOP) * (args.size / 512) ** 2
        out = {
            "metric": "SD1.5 512px 50-step images/sec (node)", "value": round(value, 4), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"SD1.5 txt2img {args.size}x{args.size}, {args.inference_steps} steps DPM++2M "
                                   f"({evals} UNet evals, CFG 7.5 parallel), batch={B} per GPU, bf16 on MI355X "
                                   f"(BASELINE.json configs[1])",
                       "images_per_step_per_gpu": B, "parallelism": f"dp{world}",
                       "weights": "random-init SD1.5 architecture (859.5 M UNet, 83.7 M VAE, 123 M CLIP)"},
            "latency_p50_s": round(statistics.median(step_times), 4),
            "latency_note": "wall time of one batch-of-8 request on rank 0 (per-image latency at batch 8)",
            "step_mfma_frac": round(alg_tflop_per_step / (elapsed / args.steps) / MFMA_PEAK_TFLOPS, 4),
            "roofline": roof,
        }
        if args.profile_all:
            out["kernel_classes"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 2),
                                         "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
                                         "gbps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)} for k, v in prof.items()}
        if world == 1 and not args.no_cpu_baseline:
            # torch's default intra-op thread count (oversubscribing all logical CPUs is several x slower)
            out["cpu_baseline"] = cpu_baseline(torch.get_num_threads())
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
