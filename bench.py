#!/usr/bin/env python
"""Headline benchmark: SD1.5 512x512, 50-step DPM++2M (51 UNet evals, CFG 7.5) images/sec on
N MI355X of one node (BASELINE.json metric; config 2: batch 8 per GPU, bf16 compute).

  python bench.py --gpus N --steps K --warmup W
  (N>1: either `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` or the plain line
   above, which then starts its own N ranks through the same launcher; fewer than N GPUs on the node = exit code 2)

A "step" is one pass of the whole hot path over one batch of synthetic input: CLIP text encode ->
51 CFG UNet evaluations (batch 16) driven by the DPM-Solver++(2M) sampler -> VAE decode of the 8
images, all inputs resident in HBM.  Rank 0 prints ONE JSON line.

--scaling weak (default): every rank generates its own 8 images; the only collective is the all_gather of the finished
    latents over RCCL.  value = all images of all ranks / max-over-ranks time.
--scaling strong: ONE request of --batch images is split over the ranks with the reference's batched_seeds rule
    (gyre_amd.sharding.shard_bounds <- services/generate.py:977-990); value = that request's images / time and
    latency_p50_s = the latency of the request (the metric's "p50 latency at 1/2/4/8").
--config sd15 (default, BASELINE configs[1]) | sdxl (configs[3]: SDXL-base topology, 1024x1024, 30 steps, 2 images per
    GPU, both text towers random-init at their real sizes) | inpaint768 (configs[2]: 9-channel SD1.5 UNet grafted onto the base UNet, 768x768,
    VAE encode of the init image, 4 images) - the extra configs are for builder / judge runs, the driver uses the default.

roofline: the MFMA kernel classes (8-wave GEMM/conv tiles, attention) are timed live with HIP events on the launch stream
during the timed region (gyre_prof_* in the C ABI); the class with the largest total is reported: achieved = algorithmic
FLOPs of its launches / their summed duration, against the 2.5 PFLOP/s dense bf16 MFMA peak; traffic = PMC-derived HBM
bytes per launch from profiles/traffic.json.  kernel_classes / roofline_hbm come from ONE extra, fully instrumented step
run AFTER the timed region (every launch bracketed by events costs a few percent, so it is kept out of `value`).
cpu_baseline: the fp32 oracle (same ATen CPU ops as the reference's CPU path) on the host cores: 1 warm-up + median of 3
CFG UNet evaluations (= one step of BASELINE configs[0]: batch 1, 512x512) and one VAE decode, combined into the
20-step Euler-a image of configs[0] and the 51-evaluation image of configs[1].
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
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)
# kernel classes timed live during the timed region; the one with the largest total is reported as the dominant kernel
CANDIDATES = ["k_gemm8<", "k_gemm4s<", "k_attn", "k_xattn"]
HBM_CLASSES = ["k_gn_", "k_layernorm"]


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


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class ClockSampler:
    """Shader clock and socket power from rocm-smi, about once per second, on a host thread - informational only: the timed region
    runs power-limited on this part (profiles/README.md), and the 2.5 PFLOP/s the fractions are priced against assumes 2.4 GHz."""

    def __init__(self):
        import re, shutil, subprocess, threading
        self._re, self._sp, self._exe = re, subprocess, shutil.which("rocm-smi")
        self.samples, self._stop, self._th = [], threading.Event(), None
        if self._exe:
            self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                o = self._sp.run([self._exe, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
                c = self._re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
                w = self._re.search(r"Power \(W\): ([\d.]+)", o)
                if c and w:
                    self.samples.append((int(c.group(1)), float(w.group(1))))
            except Exception:      # noqa: BLE001 - a monitoring hiccup must never touch the measurement
                pass
            self._stop.wait(1.0)

    def start(self):
        if self._th:
            self._th.start()

    def stop(self):
        if not self._th:
            return None
        self._stop.set()
        self._th.join(timeout=15)
        s = [x for x in self.samples if x[0] > 500]          # samples taken while the GPU was busy
        if len(s) < 2:
            return None
        clk, pw = statistics.median(a for a, _ in s), statistics.median(b for _, b in s)
        return {"sclk_mhz_median": clk, "socket_w_median": pw, "samples": len(s),
                "mfma_peak_at_that_clock_tflops": round(MFMA_PEAK_TFLOPS * clk / 2400.0, 1),
                "note": "rocm-smi about once per second during the timed region (first GPU of the box): informational - `roofline.peak` "
                        "and step_mfma_frac stay priced against 2.5 PFLOP/s = 2.4 GHz"}


def cpu_baseline():
    """Bounded CPU sample of the same workload with the oracle (kind = "port"): 1 warm-up + median of 3."""
    from gyre_amd import config as gcfg, weights
    from oracle import models_ref as M
    threads = torch.get_num_threads()          # torch's default intra-op pool (all logical CPUs oversubscribed: 10x slower)
    ucfg, vcfg = gcfg.sd15_unet(), gcfg.sd15_vae()
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg, encoder=False))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 64, 64, generator=g)
    ctx = torch.randn(2, 77, 768, generator=g)
    t = torch.tensor([981, 981])
    ts = []
    with torch.no_grad():
        for i in range(4):                      # first one is the warm-up
            t0 = time.perf_counter()
            M.unet_forward(usd, ucfg, x, t, ctx)
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        M.vae_decode(vsd, vcfg, x[:1])
        t_vae = time.perf_counter() - t0
    t_unet = statistics.median(ts[1:])
    c1, c2 = 20 * t_unet + t_vae, 51 * t_unet + t_vae
    return {"value": 1.0 / c2, "unit": "images/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model_name(), "logical_cpus": os.cpu_count(), "threads": threads,
            "unet_cfg_eval_s": {"warmup": round(ts[0], 3), "runs": [round(v, 3) for v in ts[1:]], "median": round(t_unet, 3)},
            "vae_decode_s": round(t_vae, 3),
            "config1_seconds_per_image": round(c1, 1), "config2_seconds_per_image": round(c2, 1),
            "sample": f"fp32 oracle on {threads} threads: 1 warm-up + median of 3 CFG UNet evals (batch 2, 64x64 latents = one "
                      f"step of BASELINE configs[0]) = {t_unet:.2f} s, 1 VAE decode (512x512) = {t_vae:.2f} s; configs[0] "
                      f"(20 Euler-a steps, batch 1) = 20 evals + decode = {c1:.0f} s/image, configs[1] = 51 evals + decode = "
                      f"{c2:.0f} s/image (value)"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec through torch.distributed.run with one rank per GPU on
    127.0.0.1 and hand its exit code back.  Fails loudly when the node has fewer than N GPUs (BENCH_FORCE_DEVICE, the
    two-ranks-on-one-GPU test hook, lifts that check)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and "BENCH_FORCE_DEVICE" not in os.environ:
        print(f"bench.py: --gpus {n} asked for but this node shows {have} GPU(s); one rank per GPU is required "
              f"(no rank was started, nothing was measured)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (weak) / per request (strong)")
    ap.add_argument("--inference-steps", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--config", choices=["sd15", "sdxl", "inpaint768", "tomeclip"], default="sd15")
    ap.add_argument("--tome-r", type=int, default=1024, help="tomeclip: keys / values merged per self-attention (clipped to N/2)")
    ap.add_argument("--clip-scale", type=float, default=0.2, help="tomeclip: clip_guidance_scale")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16",
                    help="16-bit storage type of activations / weights (fp32 accumulation in both): bf16 = libgyre_hip.so (default), "
                         "fp16 = libgyre_hip_f16.so, the reference's own GPU arithmetic")
    ap.add_argument("--ablation", type=lambda v: int(v, 0), default=0,
                    help="gyre_debug_gemm_ablation bits for same-box A/B runs of the whole step (0x80 = shortcuts as their own launches, "
                         "0x2000 = three-launch cross-attention; include/gyre_hip.h lists the rest); printed in the line when non-zero")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-debug", action="store_true", help="dev: torch.cuda.set_sync_debug_mode('warn') over the timed region - every "
                    "host-device synchronisation of a step is reported on stderr with the line that caused it")
    ap.add_argument("--trace-markers", action="store_true", help="dev: bracket the timed region with a fill<complex128> launch, the marker "
                    "tools/c5_trace.py / tools/trace_sequence.py cut a rocprofv3 kernel trace at")
    ap.add_argument("--no-class-table", action="store_true", help="skip the extra instrumented step after the timed region")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class INSIDE the timed region (adds event overhead)")
    args = ap.parse_args()
    defaults = {"sd15": (8, 50, 512), "sdxl": (2, 30, 1024), "inpaint768": (4, 50, 768), "tomeclip": (8, 50, 512)}[args.config]
    B = args.batch or defaults[0]
    n_steps = args.inference_steps or defaults[1]
    size = args.size or defaults[2]

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: one rank per GPU is the only form of the N-GPU path, so launch it here
        # (the same line the driver uses for its scaling run) instead of quietly measuring one GPU
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or drop the torchrun wrapper: "
                         f"`python bench.py --gpus N` starts its own ranks)")
    if "BENCH_FORCE_DEVICE" not in os.environ and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but this node shows {torch.cuda.device_count()} GPU(s): refusing to put two ranks on one device")
    # test hooks (tests/test_gpu_bench_ranks.py runs two ranks on ONE GPU, which RCCL refuses): BENCH_FORCE_DEVICE pins the
    # device index, BENCH_DIST_BACKEND=gloo swaps the collective backend.  The driver sets neither.
    dev_index = int(os.environ.get("BENCH_FORCE_DEVICE", local_rank))
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # self-check of the launch (rank 0 prints it): the world size AS THE COLLECTIVE BACKEND reports it - one all_reduce of ones over
    # the real backend (RCCL on a GPU node) - and the device every rank sits on, so that a SCALE run shows at a glance that N
    # distinct GPUs took part
    ranks_seen, rank_devices = 1, [dev_index]
    if world > 1:
        one = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        gathered = [None] * world
        try:
            pci = torch.cuda.get_device_properties(dev_index).pci_bus_id
        except Exception:
            pci = -1
        dist.all_gather_object(gathered, (rank, dev_index, pci))
        rank_devices = [list(g) for g in gathered]

    from gyre_amd import _lib, config as gcfg
    from gyre_amd.modules import GyreHipUNet, GyreHipVAE
    from gyre_amd.pipeline import GyrePipeline
    HDT = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    _lib.set_default_storage(_lib.F16 if args.dtype == "fp16" else _lib.BF16)     # the profiler / redo-counter calls below follow it
    if args.ablation:
        _lib.lib().gyre_debug_gemm_ablation(args.ablation)
    from gyre_amd.sharding import gather_batches, shard_bounds
    from gyre_amd.text import ClipTextEncoder, empty_prompt_ids, synthetic_prompt_ids

    ucfg = gcfg.sdxl_unet() if args.config == "sdxl" else gcfg.sd15_unet()
    vcfg = gcfg.sdxl_vae() if args.config == "sdxl" else gcfg.sd15_vae()
    unet = GyreHipUNet(ucfg).to(HDT).to(dev)
    fill_synthetic_on_device(unet, 0)
    vae = GyreHipVAE(vcfg).to(HDT).to(dev)
    fill_synthetic_on_device(vae, 1)
    inpaint = None
    if args.config == "inpaint768":
        inpaint = GyreHipUNet(gcfg.sd15_unet(in_channels=9)).to(HDT).to(dev)
        fill_synthetic_on_device(inpaint, 3)
    clip = None if args.config == "sdxl" else ClipTextEncoder.synthetic(dev, HDT, seed=2)
    sdxl_cond = None
    if args.config == "sdxl":
        # both SDXL text towers at their real sizes (CLIP ViT-L/14 and OpenCLIP ViT-bigG/14 text models, random init), run on the
        # host path inside every step as the SD1.x CLIP encode is (gyre_amd/text.py SDXLTextConditioner)
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
        from gyre_amd.text import SDXLTextConditioner, sdxl_time_ids
        kw_ = dict(vocab_size=49408, max_position_embeddings=77, bos_token_id=49406, eos_token_id=49407, pad_token_id=49407)
        torch.manual_seed(5)
        with torch.device(dev):
            te1 = CLIPTextModel(CLIPTextConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                                               hidden_act="quick_gelu", **kw_))
            te2 = CLIPTextModelWithProjection(CLIPTextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32,
                                                             num_attention_heads=20, hidden_act="gelu", projection_dim=1280, **kw_))
        te1, te2 = te1.to(HDT).eval(), te2.to(HDT).eval()
        ident = lambda frag: frag                       # prompts arrive pre-tokenised: [(token ids, weight)]
        sdxl_cond = SDXLTextConditioner(te1, ident, te2, ident, dev)
    clip_model = fe = None
    if args.config == "tomeclip":
        # BASELINE configs[4]: ToMe + CLIP guidance.  The CLIP model is host PyTorch by north_star: a random-init ViT-B/32
        # (transformers' default CLIPConfig sizes, 224 px) - no checkpoint exists offline
        from types import SimpleNamespace
        from transformers import CLIPConfig, CLIPModel
        from gyre_amd.clipguided import patch_embedding_as_matmul
        torch.manual_seed(7)
        # bf16 like the rest of the pipeline (the reference loads clip_model in the engine's fp16): the guidance follows the dtype the
        # caller loaded the model in (gyre_amd/clipguided.py cond_fn); fp32 here put ~10 % of the step into rocBLAS fp32 GEMMs
        clip_model = patch_embedding_as_matmul(CLIPModel(CLIPConfig(projection_dim=512)).eval().to(dev).to(HDT))
        for p_ in clip_model.parameters():
            p_.requires_grad_(False)
        fe = SimpleNamespace(image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711],
                             size={"shortest_edge": 224})
        unet.set_tome(args.tome_r)
    pipe = GyrePipeline(unet, vae, clip, device=dev, inpaint_unet=inpaint, grafted_inpaint=inpaint is not None,
                        clip_model=clip_model, feature_extractor=fe)

    # strong scaling: one request of B images over the ranks; weak: B images on every rank
    if args.scaling == "strong":
        lo, hi = shard_bounds(B, world)[rank]
        sizes = [e - s for s, e in shard_bounds(B, world)]
    else:
        lo, hi = 0, B
        sizes = [B] * world
    nloc = hi - lo
    pr = 0 if args.scaling == "strong" else rank           # strong: every rank sees the same request
    extra = {}
    if args.config == "sdxl":
        sd_ids = synthetic_prompt_ids(B, seed=1234 + pr)
        sd_prompts = [[([int(t) for t in row[1:] if int(t) != 49407], 1.0)] for row in sd_ids]     # token ids without BOS / EOS padding
        extra = dict(guidance_scale=5.0)
    else:
        ids = synthetic_prompt_ids(B, seed=1234 + pr).to(dev)
        neg = empty_prompt_ids(B).to(dev)
    if args.config == "tomeclip":
        # gradient_threshold 0: the flat-loss stop never triggers, every step of every request is guided (worst case)
        extra = dict(clip_guidance_scale=args.clip_scale, clip_input_ids=synthetic_prompt_ids(B, seed=99 + pr).to(dev)[lo:hi],
                     clip_gradient_threshold=0.0)
    if args.config == "inpaint768":
        yy, xx = torch.meshgrid(torch.linspace(0, 1, size), torch.linspace(0, 1, size), indexing="ij")
        init = torch.stack([yy, xx, (yy + xx) / 2])[None].to(dev)
        mask = torch.zeros(1, 1, size, size, device=dev)
        mask[:, :, size // 4: 3 * size // 4, size // 4: 3 * size // 4] = 1.0
        extra = dict(image=init, mask_image=mask, strength=1.0)

    def step(i):
        if nloc == 0:
            latents = torch.zeros((0, 4, size // 8, size // 8), device=dev)
            images = None
        else:
            base = 420420420 + (0 if args.scaling == "strong" else rank * 100000) + i * B
            seeds = [base + j for j in range(lo, hi)]
            if args.config == "sdxl":
                cond_, pooled_, unc_, upooled_ = sdxl_cond(sd_prompts[lo:hi], [""] * nloc, True)
                tids = sdxl_time_ids(nloc, size, size, device=dev)
                kw = dict(text_embeddings=cond_, uncond_embeddings=unc_, added_cond={"text_embeds": pooled_, "time_ids": tids},
                          uncond_added_cond={"text_embeds": upooled_, "time_ids": tids})
            else:
                kw = dict(input_ids=ids[lo:hi], negative_ids=neg[lo:hi])
            # per-image generators: on the execution device like the reference's wrapper (pipeline_wrapper.py:246) where the per-step
            # draws are TENSORS (the graft's blend maps of config 3: a CPU generator costs a synchronous host-to-device copy per draw);
            # on the host where the draws are SCALARS the host needs at once (the cut-out sizes of config 5: a device generator costs
            # a device-to-host read-back per cut-out).  DPM++2M itself draws nothing per step.
            gen_dev = str(dev) if args.config == "inpaint768" else "cpu"
            latents = pipe(seeds=seeds, height=size, width=size, num_inference_steps=n_steps, generator_device=gen_dev,
                           **{"guidance_scale": 7.5, "sampler": "dpmpp_2m", "output_type": "latent", **extra, **kw})
            images = pipe.vae_decode(latents)
        if world > 1:
            latents = gather_batches(latents, sizes)  # RCCL all_gather of the finished latents
        return images, latents

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up.  The last warm-up step runs with every MFMA kernel family instrumented to find the DOMINANT class; the timed
    # region then carries HIP events around that class's launches only (events around all ~16 k launches of a step cost ~7 %:
    # 0.99 s vs 1.07 s per request on the same box - that overhead belongs to the measurement, not to `value`).
    dom_prefix = None
    for i in range(args.warmup):
        last = i == args.warmup - 1
        if last and not args.profile_all:
            _lib.prof_enable(CANDIDATES)
        step(-1 - i)
        if last and not args.profile_all:
            torch.cuda.synchronize()
            pre = {k: v for k, v in _lib.prof_collect().items() if any(k.startswith(c) for c in CANDIDATES)}
            _lib.prof_enable([])
            if pre:
                dom_prefix = max(pre, key=lambda k: pre[k]["ms"])
    barrier()
    _lib.prof_enable(None if args.profile_all else ([dom_prefix] if dom_prefix else CANDIDATES))
    step_times = []
    clocks = ClockSampler() if rank == 0 else None
    if clocks:
        clocks.start()
    redo0 = _lib.lib().gyre_debug_attn_redo_count()
    if args.trace_markers:
        torch.full((1,), 1.0, dtype=torch.complex128, device=dev)
    if args.sync_debug:
        import traceback, warnings
        def _show(message, category, filename, lineno, file=None, line=None):
            if "synchroniz" in str(message):
                fr = [f for f in traceback.extract_stack()[:-2] if "/torch/" not in f.filename and "warnings" not in f.filename]
                sys.stderr.write("[sync] " + " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-4:][::-1]) + "\n")
        warnings.showwarning = _show
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
    t0 = time.perf_counter()
    for i in range(args.steps):
        s0 = time.perf_counter()
        images, latents = step(i)
        torch.cuda.synchronize()
        step_times.append(time.perf_counter() - s0)
    if args.sync_debug:
        torch.cuda.set_sync_debug_mode("default")
    barrier()
    elapsed = time.perf_counter() - t0
    if args.trace_markers:
        torch.full((1,), 1.0, dtype=torch.complex128, device=dev)
        torch.cuda.synchronize()
    clock_info = clocks.stop() if clocks else None
    redo1 = _lib.lib().gyre_debug_attn_redo_count()
    prof = _lib.prof_collect()
    _lib.prof_enable([])
    if images is not None:
        assert bool(torch.isfinite(images).all()), "non-finite output"
    evals = getattr(pipe, "last_unet_evals", 0)

    # ---- the latency half of the metric (after the timed region; never part of `value`) -------------------------------------
    # (a) one image alone: UNet batch 2 (CFG), the per-image latency floor of a single GPU
    # (b) ONE batch-of-B request split over all ranks with the reference's batched_seeds rule (= --scaling strong): what a
    #     Gyre client sees when the node's GPUs serve its request together.  Every rank takes part (all_gather of the latents).
    latency = {}
    if args.config == "sd15" and args.scaling == "weak":
        def one_request(rlo, rhi, total):
            n = rhi - rlo
            if n > 0:
                seeds_ = [420420420 + j for j in range(rlo, rhi)]
                lat_ = pipe(seeds=seeds_, height=size, width=size, num_inference_steps=n_steps, guidance_scale=7.5,
                            sampler="dpmpp_2m", output_type="latent", input_ids=ids0[rlo:rhi], negative_ids=neg[rlo:rhi])
                pipe.vae_decode(lat_)
            else:
                lat_ = torch.zeros((0, 4, size // 8, size // 8), device=dev)
            if world > 1:
                lat_ = gather_batches(lat_, [e - s_ for s_, e in shard_bounds(total, world)])
            return lat_

        def timed(fn, reps):
            ts_ = []
            for _ in range(reps):
                barrier()
                t_ = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts_.append(time.perf_counter() - t_)
            tt_ = torch.tensor(ts_, device=dev if (world > 1 and backend == "nccl") else "cpu", dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            return float(tt_.median())
        ids0 = synthetic_prompt_ids(B, seed=1234).to(dev)          # the SAME request on every rank
        if world == 1:
            one_request(0, 1, 1)                                    # warm-up at UNet batch 2 (workspace, planner)
            latency["latency_b1_s"] = round(timed(lambda: one_request(0, 1, 1), 2), 4)
            latency["latency_b1_note"] = "wall time of a ONE-image request on one GPU (51 CFG UNet evaluations at batch 2 + VAE decode)"
        slo, shi = shard_bounds(B, world)[rank]
        one_request(slo, shi, B)                                    # warm-up at the shard's batch size
        latency["latency_request_split_s"] = round(timed(lambda: one_request(slo, shi, B), 2), 4)
        latency["latency_request_split_note"] = (f"wall time of ONE batch-of-{B} request split over the {world} GPU(s) of the node "
                                                 f"(batched_seeds rule, per-rank images {[e - s_ for s_, e in shard_bounds(B, world)]}; "
                                                 f"max over ranks, median of 2)")
    if world > 1:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- the same steps with the attention kernel's per-tile overflow check in EVERY tile (variant 7), after the timed region:
    # the default pass is optimistic (a workgroup whose row sums leave (0, 2^60) repeats its tile with the checked pass - bit-identical
    # results either way), so on data whose logits trip the bound the default can cost up to checked + optimistic; this is the
    # data-INDEPENDENT figure, `attn_redo_count` says how far the timed region was from it (0 = no workgroup repeated anything)
    checked = None
    if args.config == "sd15":
        n_chk = min(2, args.steps)
        old_variant = _lib.lib().gyre_debug_force_attn_variant(7)
        try:
            step(args.steps + 1)                                    # warm-up (nothing to plan: same shapes, same workspace)
            barrier()
            c0 = time.perf_counter()
            for i in range(n_chk):
                step(args.steps + 2 + i)
            barrier()
            c_chk = time.perf_counter() - c0
        finally:
            _lib.lib().gyre_debug_force_attn_variant(old_variant)
        if world > 1:
            tt = torch.tensor([c_chk], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            c_chk = float(tt.item())
        checked = (sum(sizes) * n_chk / c_chk, n_chk)

    # one extra, fully instrumented step (outside the timed region): the per-class table
    classes = None
    if not args.no_class_table and not args.profile_all:
        # EVERY rank runs the extra step (it contains the all_gather of the finished latents - a collective that only rank 0
        # entered would hang the job); only rank 0 instruments it
        if rank == 0:
            _lib.prof_enable(None)
        c0 = time.perf_counter()
        step(args.steps)
        torch.cuda.synchronize()
        c_el = time.perf_counter() - c0
        if rank == 0:
            classes = _lib.prof_collect() if nloc else None
            _lib.prof_enable([])
    elif args.profile_all:
        classes, c_el = prof, elapsed
    if world > 1:
        dist.barrier()

    if rank == 0:
        total_images = sum(sizes) * args.steps
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
                    "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                    "share_of_step_time": round(d["ms"] * 1e-3 / elapsed, 4),
                    "traffic_note": "traffic = class mean of 2*FETCH_SIZE + WRITE_SIZE (profiles/traffic.json, separate rocprofv3 --pmc "
                                    "passes): L2 <- fabric requests of all eight XCD L2s incl. each one's own copy of the weight slice and "
                                    "the fp32 split-K slabs; per shape the 64x64 convs move 1.18-1.23x their algorithmic bytes "
                                    "(profiles/README.md)"}
        per_img = {"sd15": evals * 2 * UNET_TFLOP_PER_SAMPLE * (size / 512) ** 2 + VAE_DEC_TFLOP * (size / 512) ** 2}.get(args.config)
        out = {
            "metric": {"sd15": "SD1.5 512px 50-step images/sec (node)", "sdxl": "SDXL-base 1024px 30-step images/sec (node)",
                       "inpaint768": "SD1.5 grafted inpaint 768px images/sec (node)",
                       "tomeclip": "SD1.5 512px 50-step ToMe + CLIP-guided images/sec (node)"}[args.config],
            "value": round(value, 4), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": {
                "sd15": f"SD1.5 txt2img {size}x{size}, {n_steps} steps DPM++2M ({evals} UNet evals, CFG 7.5 parallel), "
                        f"batch={B} per {'request' if args.scaling == 'strong' else 'GPU'}, {args.dtype} on MI355X (BASELINE.json configs[1])",
                "sdxl": f"SDXL-base topology txt2img {size}x{size}, {n_steps} steps DPM++2M ({evals} UNet evals, CFG 5), batch={B} per "
                        f"{'request' if args.scaling == 'strong' else 'GPU'}, {args.dtype}, both text towers (random-init CLIP ViT-L + OpenCLIP bigG text models) on the host path (BASELINE.json configs[3]; not in the reference)",
                "inpaint768": f"SD1.5 grafted inpaint {size}x{size} (9-ch inpaint UNet + base UNet, hires fix, VAE encode), {n_steps} steps "
                              f"DPM++2M ({evals} UNet evals), batch={B}, {args.dtype} (BASELINE.json configs[2])",
                "tomeclip": f"SD1.5 txt2img {size}x{size}, ToMe r={args.tome_r} + CLIP guidance (scale {args.clip_scale}, guided base, 2 + 2 "
                            f"cut-outs, every step guided; random-init ViT-B/32 in host PyTorch, bf16), {n_steps} steps DPM++2M ({evals} sampler-level "
                            f"UNet evals incl. the differentiated stems: per guided step ONE activation-keeping native pass over cat[uncond, cond] + the "
                            f"reverse sweep of its conditional half), batch={B}, {args.dtype} (BASELINE.json configs[4])"}[args.config],
                       "images_per_step": sum(sizes), "images_per_rank": sizes, "parallelism": f"dp{world}",
                       "dist_backend": backend if world > 1 else None, "rccl_ranks_seen": ranks_seen,
                       "rank_devices": rank_devices,
                       "weights": "random-init weights of the exact architecture (SD1.5: 859.5 M UNet, 83.7 M VAE, 123 M CLIP)"},
            "latency_p50_s": round(statistics.median(step_times), 4),
            "latency_note": ("wall time of ONE request of %d images split over %d GPUs (rank 0)" % (B, world)) if args.scaling == "strong"
                            else "wall time of one batch-of-%d request on rank 0 (per-image latency at batch %d)" % (B, B),
            "roofline": roof,
        }
        out.update(latency)
        if args.ablation:
            out["ablation_bits"] = hex(args.ablation)
        out["attn_redo_count"] = (redo1 - redo0) if (redo0 >= 0 and redo1 >= 0) else None
        out["attn_redo_note"] = ("attention workgroups of rank 0's device that repeated their tile with the checked pass during the timed "
                                 "region (gyre_debug_attn_redo_count; of ~%d k_attn3 workgroups per step)" % (evals * 5 * 4096 if args.config == "sd15" else 0))
        if checked:
            out["value_checked_attention"] = round(checked[0], 4)
            out["value_checked_attention_note"] = (f"images/s of {checked[1]} further step(s) with the per-tile overflow check in every attention tile "
                                                   f"(gyre_debug_force_attn_variant(7)): the data-independent form; outputs are bit-identical to the default's")
        if clock_info and clock_info.get("sclk_mhz_median"):
            out["images_per_s_per_ghz"] = round(value / (clock_info["sclk_mhz_median"] / 1000.0), 3)
        if per_img:
            # CFG-parallel calls share the part of the network in front of the first cross-attention between the two halves of
            # the batch (gyre_unet_hint_cfg_pairs): conv_in, the first resnet, proj_in, Q|K|V, the 64x64 self-attention and its
            # to_out = 655.3 of the 12 848 GFLOP of a batch-16 evaluation, executed once per pair -> 2.55 % of the reference
            # formulation's FLOPs are not executed.  step_mfma_frac counts EXECUTED FLOPs only.
            shared = args.config == "sd15" and os.environ.get("GYRE_CFG_SHARED_PREFIX", "1") != "0"
            executed = 1.0 - (0.0255 if shared else 0.0)
            alg = sum(sizes) * per_img / (elapsed / args.steps) / MFMA_PEAK_TFLOPS / world
            unet_part = evals * 2 * UNET_TFLOP_PER_SAMPLE * (size / 512) ** 2 / per_img
            out["step_mfma_frac"] = round(alg * (1.0 - unet_part * (1.0 - executed)), 4)
            out["step_mfma_frac_reference_formulation"] = round(alg, 4)
            out["cfg_shared_prefix"] = shared
            out["step_mfma_frac_note"] = ("EXECUTED FLOPs / time / 2.5 PFLOP/s: algorithmic 0.803 TFLOP per UNet sample-forward x all "
                                          "evaluations, minus what the two halves of every CFG-parallel call share (everything in front "
                                          "of the first cross-attention is evaluated once per (uncond, cond) pair: 2.55 % of a call; "
                                          "identical between the two halves, equal to the unshared call up to bf16 summation order (bit for bit under batch-invariant planning), GYRE_CFG_SHARED_PREFIX=0 turns it off; "
                                          "step_mfma_frac_reference_formulation counts those FLOPs as if executed twice); the "
                                          "cross-attention K/V projections of the text context run once per request (context cache: "
                                          "<0.5 % of the counted FLOPs are not executed on 50 of the 51 evaluations)")
        if classes:
            tot = sum(v["ms"] for v in classes.values())
            table = {}
            for k, v in sorted(classes.items(), key=lambda kv: -kv[1]["ms"]):
                tf = v["flops"] / max(v["ms"], 1e-9) / 1e9
                gb = v["bytes"] / max(v["ms"], 1e-9) / 1e6
                table[k] = {"launches": v["launches"], "ms": round(v["ms"], 2), "share": round(v["ms"] / tot, 4),
                            "tflops": round(tf, 1), "mfma_frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                            "gbps": round(gb, 1), "hbm_frac": round(gb / HBM_PEAK_GBPS, 4)}
            out["kernel_classes"] = table
            fams = {"k_gemm8 (non-pipelined tile GEMMs / convs)": ("k_gemm8<",), "k_gemm4s (pipelined convs / long-K GEMMs)": ("k_gemm4s<",),
                    "k_attn": ("k_attn",), "k_xattn (fused cross-attention block)": ("k_xattn",), "k_gemm_ar": ("k_gemm_ar",), "k_gemm_sm + 4-wave k_gemm": ("k_gemm_sm", "k_gemm<"),
                    "GroupNorm (k_gn_*)": ("k_gn_",), "split-K reduction": ("k_splitk",), "LayerNorm": ("k_layernorm",)}
            fam = {}
            for fname, prefixes in fams.items():
                sel = [v for k, v in classes.items() if any(k.startswith(p_) for p_ in prefixes)]
                if sel:
                    ms_ = sum(v["ms"] for v in sel)
                    tf = sum(v["flops"] for v in sel) / max(ms_, 1e-9) / 1e9
                    gb = sum(v["bytes"] for v in sel) / max(ms_, 1e-9) / 1e6
                    fam[fname] = {"launches": sum(v["launches"] for v in sel), "ms": round(ms_, 2), "share": round(ms_ / tot, 4),
                                  "mfma_frac": round(tf / MFMA_PEAK_TFLOPS, 4), "hbm_frac": round(gb / HBM_PEAK_GBPS, 4)}
            fam["GroupNorm + split-K share"] = round(sum(v["ms"] for k, v in classes.items() if k.startswith(("k_gn_", "k_splitk"))) / tot, 4)
            out["kernel_families"] = fam
            out["kernel_classes_note"] = (f"HIP-event time per kernel class in one extra fully instrumented step after the timed region "
                                          f"({c_el * 1e3:.0f} ms wall, sum of classes {tot:.0f} ms); tflops / gbps = algorithmic FLOPs / bytes "
                                          f"of the unpadded problems over event-to-event time")
            hb = {k: v for k, v in classes.items() if any(k.startswith(c) for c in HBM_CLASSES)}
            if hb:
                k = max(hb, key=lambda n: hb[n]["ms"])
                gb = hb[k]["bytes"] / hb[k]["ms"] / 1e6
                out["roofline_hbm"] = {"bound": "hbm", "kernel": k, "achieved": round(gb, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": round(gb / HBM_PEAK_GBPS, 4), "launches": hb[k]["launches"],
                                       "avg_launch_us": round(hb[k]["ms"] * 1e3 / hb[k]["launches"], 2),
                                       "share_of_step_time": round(hb[k]["ms"] / tot, 4)}
        if clock_info:
            out["clock_power"] = clock_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
