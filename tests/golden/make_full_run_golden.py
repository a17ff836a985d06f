#!/usr/bin/env python
"""Generate tests/golden/full_run_latents.npz: the fp32 CPU ORACLE's final latents of the two full-length
headline runs SURVEY.md 8(d) gates on ("c1 / c2 final image PSNR >= 30 dB, state the achieved number").

  c1  BASELINE configs[0]: SD1.5 512x512, 20 steps euler_a, CFG 7.5, batch 1, seed 420420420
      (the reference's tests/happy_path.py request shape; sampler selected at samplers.py:50)
  c2  BASELINE configs[1]: SD1.5 512x512, 50 steps DPM++2M with the LMS warm-up (51 UNet evaluations,
      schedulers/sample_dpmpp_2m.py:6-50), CFG 7.5, 2 of the batch's images (seeds 420420420, 420420421;
      images of a batch never interact, randtools.py:39-64, so 2 images pin the per-image trajectory)

The oracle is oracle/pipeline_ref.generate_ref over seeded synthetic weights of the exact SD1.5 architecture
(gyre_amd.weights.synthetic_state_dict; no checkpoint exists offline).  Nothing of /root/reference is read:
this is the build's OWN restatement run to full length on the CPU (UNet / VAE leaves stay parity-unpinned,
DESIGN.md 2).  It only needs CPU time (about 45 min on 8 cores), which is why it is a committed fixture and
not a leg of the GPU test: tests/test_gpu_full_runs.py replays the same requests on the HIP path and reports
latent rel-L2 and image PSNR against these tensors.

Run:  python tests/golden/make_full_run_golden.py [c1] [c2]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gyre_amd import config as gcfg, weights  # noqa: E402
from oracle import pipeline_ref as PR  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "full_run_latents.npz")

RUNS = {
    "c1": dict(seeds=[420420420], steps=20, sampler="euler_a", emb_seed=9),
    "c2": dict(seeds=[420420420, 420420421], steps=50, sampler="dpmpp_2m", emb_seed=11),
}


def embeddings(run):
    """Synthetic text / negative embeddings of a run (shared with the GPU test)."""
    g = torch.Generator().manual_seed(run["emb_seed"])
    n = len(run["seeds"])
    text = torch.randn(n, 77, 768, generator=g)
    unc = torch.randn(1, 77, 768, generator=g).expand(n, -1, -1).contiguous()
    return text, unc


def main():
    which = [a for a in sys.argv[1:] if a in RUNS] or list(RUNS)
    ucfg, vcfg = gcfg.sd15_unet(), gcfg.sd15_vae()
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg))
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for name in which:
        run = RUNS[name]
        text, unc = embeddings(run)
        t0 = time.time()
        with torch.no_grad():
            lat, evals = PR.generate_ref(usd, ucfg, vsd, vcfg, text, unc, run["seeds"], 512, 512, run["steps"], 7.5,
                                         run["sampler"], decode=False)
        print(f"{name}: {evals} UNet evaluations, {time.time() - t0:.0f} s, latents {tuple(lat.shape)} "
              f"absmax {float(lat.abs().max()):.3f} std {float(lat.std()):.3f}", flush=True)
        out[name + "_latents"] = lat.numpy().astype(np.float32)
        out[name + "_evals"] = np.int64(evals)
        np.savez_compressed(OUT, **out)


if __name__ == "__main__":
    main()
