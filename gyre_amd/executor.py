"""One request, several device slots, ONE process: the in-server form of the data-parallel sharding.

north_star asks for a request batch split over the GPUs of a node; the reference serves its devices from the threads of a
single process - a ``DeviceQueueSlot`` per GPU, one request per slot at a time (gyre/manager.py:648-651,2106-2141, gRPC
``ThreadPoolExecutor`` server.py:154) - and runs the sub-batches of one request one after the other on one slot
(services/generate.py:1049-1091).  ``gyre_amd.sharding`` is the one-process-per-GPU form (torchrun + RCCL) the bench uses;
this module is the form a Gyre server process can call: the images of a request are split with the reference's own
``batched_seeds`` rule (services/generate.py:977-990 = ``sharding.shard_bounds``) over replicas of the pipeline, every
replica runs its slice on its own device from its own host thread and HIP stream (the C ABI is safe for concurrent use of
different handles, include/gyre_hip.h; tests/test_gpu_threads.py), and the finished latents come home with one device-to-
device copy per replica (``hipMemcpyPeerAsync`` under ``Tensor.to`` over xGMI - 32 KB per image; there is no data-path
collective to run).  Weights are replicated; every random draw is per image (randtools.py:39-64), so the
result does not depend on the split - bit for bit with ``bit_exact=True`` (batch-invariant split-K planning per worker
thread), to bf16 rounding otherwise (the GEMM planner may choose another split-K factor for another batch size).

    ex = DeviceSlotExecutor.replicate(pipe, ["cuda:0", "cuda:1", ...])       # or from ready-made per-device pipelines
    latents = ex(seeds=[...], text_embeddings=te, uncond_embeddings=ue, num_inference_steps=50, sampler="dpmpp_2m")
    images = ex.decode(latents)                                              # VAE decode, sharded the same way

``GyreUnifiedPipeline`` (gyre_amd/engine.py) uses it when the engine option ``shard_devices`` names more than one device.
"""
from __future__ import annotations

import copy
import threading
from typing import List, Optional, Sequence

import torch

from .pipeline import GyrePipeline
from .sharding import shard_bounds

def normalize_device(d) -> torch.device:
    """'cuda' and 'cuda:<current>' are the same slot: give every CUDA device its explicit index."""
    d = torch.device(d)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return d


def weights_key(pipe: GyrePipeline, devices: Sequence):
    """What a set of replicas was made FROM: the identity and weight version (bumped by load_state_dict, .to()/.half(),
    _invalidate()) of every native module of ``pipe`` and the normalised device list."""
    mods = tuple(getattr(pipe, n, None) for n in ("unet", "vae", "inpaint_unet"))
    return (tuple((id(m), getattr(m, "_weights_version", 0)) for m in mods), tuple(str(normalize_device(d)) for d in devices))


_PER_IMAGE = ("text_embeddings", "uncond_embeddings", "input_ids", "negative_ids", "clip_input_ids", "clip_text_embeddings")


class DeviceSlotExecutor:
    def __init__(self, pipelines: Sequence[GyrePipeline]):
        if not pipelines:
            raise ValueError("at least one pipeline replica is required")
        self.pipelines: List[GyrePipeline] = list(pipelines)
        self.last_unet_evals = 0
        self.source: Optional[GyrePipeline] = None      # replicate(): the pipeline the replicas were copied from
        self.source_key = None                          # ... and weights_key() of it at that moment

    # ---- construction -------------------------------------------------------------------------------------------------
    @classmethod
    def replicate(cls, pipe: GyrePipeline, devices: Sequence) -> "DeviceSlotExecutor":
        """Weights replicated: ``pipe`` serves the first device (moved there if it lives elsewhere), a deep copy of its
        modules each further one - the reference's ``clone_model`` + ``.to(device)`` per slot (pipeline_wrapper.py:114-131)."""
        devs = [normalize_device(d) for d in devices]
        home = normalize_device(pipe.device)
        reps = []
        for i, d in enumerate(devs):
            if d == home and pipe not in reps:          # the source serves its own device wherever it sits in the list
                reps.append(pipe)
                continue
            clone = lambda m: None if m is None else copy.deepcopy(m).to(d)
            reps.append(GyrePipeline(clone(pipe.unet), clone(pipe.vae), clone(getattr(pipe, "text_encoder", None)), device=d,
                                     inpaint_unet=clone(getattr(pipe, "inpaint_unet", None)),
                                     grafted_inpaint=getattr(pipe, "grafted_inpaint", False),
                                     clip_model=clone(getattr(pipe, "clip_model", None)),
                                     feature_extractor=getattr(pipe, "feature_extractor", None)))
            for attr in ("hires_fix", "hires_threshold_fraction", "hires_oos_fraction", "hires_image_oos_fraction"):
                if hasattr(pipe, attr):
                    setattr(reps[-1], attr, getattr(pipe, attr))
        ex = cls(reps)
        ex.source, ex.source_key = pipe, weights_key(pipe, devs)
        return ex

    def stale(self, devices: Optional[Sequence] = None) -> bool:
        """True when the source modules changed (other objects, new weights, another dtype / device) since replicate()."""
        if self.source is None:
            return False
        devs = devices if devices is not None else [p.device for p in self.pipelines]
        return weights_key(self.source, devs) != self.source_key

    @property
    def world(self) -> int:
        return len(self.pipelines)

    # ---- one request ----------------------------------------------------------------------------------------------------
    def _fan(self, n_items: int, work):
        """work(rank, start, end) on a thread per replica with a non-empty slice; results in rank order."""
        bounds = shard_bounds(n_items, self.world)
        out: List[Optional[torch.Tensor]] = [None] * self.world
        errs: List[BaseException] = []
        # work the caller already queued on a device (e.g. a previous request on the same replica) must finish before a worker
        # stream touches that replica's workspace: every worker stream first waits for the caller's current stream there
        ahead = {}
        for pl in self.pipelines:
            d = torch.device(pl.device)
            if d.type == "cuda" and d not in ahead:
                ahead[d] = torch.cuda.current_stream(d)

        def run(r, s, e):
            try:
                dev = torch.device(self.pipelines[r].device)
                if dev.type == "cuda":
                    torch.cuda.set_device(dev)
                    stream = torch.cuda.Stream(device=dev)
                    stream.wait_stream(ahead[dev])
                    with torch.cuda.stream(stream):
                        out[r] = work(r, s, e)
                    stream.synchronize()
                else:
                    out[r] = work(r, s, e)
            except BaseException as exc:  # noqa: BLE001 - re-raised on the calling thread
                errs.append(exc)
        jobs = [(r, s, e) for r, (s, e) in enumerate(bounds) if e > s]
        if len(jobs) == 1:
            run(*jobs[0])
        else:
            threads = [threading.Thread(target=run, args=j, name=f"gyre-slot-{j[0]}") for j in jobs]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if errs:
            raise errs[0]
        return bounds, out

    def __call__(self, *, seeds: Optional[Sequence[int]] = None, generators: Optional[Sequence[torch.Generator]] = None,
                 bit_exact: bool = False, gather_device=None, **kw) -> torch.Tensor:
        """The keywords of ``GyrePipeline.__call__``; per-image tensors (embeddings, ids, SDXL conditioning) follow their
        image's slice, everything else (init image, masks, scalars) is shared.  Returns the finished latents of the WHOLE
        request on ``gather_device`` (default: the first replica's device), in request order."""
        if (seeds is None) == (generators is None):
            raise ValueError("pass either seeds or generators (one per image)")
        items = list(seeds) if seeds is not None else list(generators)
        B = len(items)
        if B < 1:
            raise ValueError("at least one seed (image) is required")
        if kw.get("callback") is not None and self.world > 1:
            raise NotImplementedError("per-step callbacks of a request that is split over device slots")
        kw = dict(kw)
        kw["output_type"] = "latent"
        home = torch.device(gather_device) if gather_device is not None else torch.device(self.pipelines[0].device)
        evals = [0] * self.world

        def work(r, s, e):
            pipe = self.pipelines[r]
            dev = torch.device(pipe.device)
            sub = {}
            for k, v in kw.items():
                if v is None:
                    sub[k] = v
                elif k in _PER_IMAGE and isinstance(v, torch.Tensor):
                    sub[k] = (v if v.shape[0] == 1 else v[s:e]).to(dev, non_blocking=True)
                elif k in ("added_cond", "uncond_added_cond") and isinstance(v, dict):
                    sub[k] = {kk: (vv if vv.shape[0] == 1 else vv[s:e]).to(dev) for kk, vv in v.items()}
                elif isinstance(v, torch.Tensor):
                    sub[k] = v.to(dev)
                else:
                    sub[k] = v
            sub["seeds" if seeds is not None else "generators"] = items[s:e]
            prev = None
            if bit_exact and dev.type == "cuda":        # thread-local planner switch (include/gyre_hip.h)
                from .modules import set_batch_invariant
                prev = set_batch_invariant(16)
            try:
                lat = pipe(**sub)
            finally:
                if prev is not None:
                    from .modules import set_batch_invariant
                    set_batch_invariant(prev)
            evals[r] = getattr(pipe, "last_unet_evals", 0)
            return lat if lat.device == home else lat.to(home, non_blocking=False)   # peer copy over xGMI
        bounds, parts = self._fan(B, work)
        self.last_unet_evals = max(evals)
        return torch.cat([p for p in parts if p is not None], dim=0)

    def decode(self, latents: torch.Tensor, gather_device=None) -> torch.Tensor:
        """VAE decode of the gathered latents, sharded like the request (2.5 TFLOP per 512x512 image shards too)."""
        home = torch.device(gather_device) if gather_device is not None else latents.device

        def work(r, s, e):
            pipe = self.pipelines[r]
            img = pipe.vae_decode(latents[s:e].to(pipe.device))
            return img if img.device == home else img.to(home)
        _, parts = self._fan(latents.shape[0], work)
        return torch.cat([p for p in parts if p is not None], dim=0)
