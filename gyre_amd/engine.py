"""Engine-level drop-in: the reference's ``UnifiedPipeline`` call contract over the native pipeline.

An existing Gyre server selects it in engines.yaml:

    - id: "stable-diffusion-v1-5-mi355x"
      class: "gyre_amd.engine.GyreUnifiedPipeline"
      model: "@sd15"
      overrides:
        unet: {class: "gyre_amd.modules.GyreHipUNet"}
        vae:  {class: "gyre_amd.modules.GyreHipVAE"}

and everything above it stays the reference's own code: ``EngineManager`` builds the object from the loaded modules
(manager.py:1712-1783: ``Class(**modules)``), wraps it in ``DiffusionPipelineWrapper`` (pipeline_wrapper.py:164-397), the
gRPC servicer (services/generate.py:992-1152) calls the wrapper with the request's kwargs and turns the returned tensors
into PNG artifacts.  What this class has to honour is therefore exactly what the wrapper touches:

  * constructor keywords = module names (vae, text_encoder, tokenizer, unet, scheduler, inpaint_unet, ...);
    ``pipeline_modules()`` so that ``PipelineWrapper.activate(device)`` can swap in per-device clones (:96-131)
  * ``scheduler`` attribute: the wrapper injects the sampler chosen by the request (:255-267) - a k-diffusion sampler
    callable or a diffusers scheduler object; it is mapped back to the native sampler of the same name
  * ``progress_bar`` attribute: wrapped around the step iterable, raises ``ProgressBarAbort`` on cancellation (:26-47)
  * ``__call__`` keywords of ``UnifiedPipeline.__call__`` (unified_pipeline.py:1722-1790) - the wrapper filters its
    kwargs by this signature (:269-286) - returning ``(images [B,3,H,W] in 0..1, nsfw flags)`` for
    ``output_type="tensor", return_dict=False`` (:2512-2534)
  * ``set_options`` with the reference's option names (unified_pipeline.py:1538-1629)
  * no-op memory knobs (attention / VAE slicing, xformers): 288 GB of HBM3E, the whole batch stays resident.

Features outside the native hot path raise NotImplementedError (-> gRPC UNIMPLEMENTED, services/exception_to_grpc.py):
depth / hint images (ControlNet, T2I), textual-inversion token embeddings.  CLIP guidance
is implemented (gyre_amd/clipguided.py over the native input-gradient sweeps).  The safety checker stays the host module the
manager loaded; it is RUN exactly as the reference runs it (``_safety_check``), never skipped silently.
"""
from __future__ import annotations

import functools
from typing import Any, Callable, List, Optional

import torch

from . import clipguided as CG
from . import lora as LR
from .pipeline import GyrePipeline
from .text import LPWTextEmbedder

# sampler callables (k-diffusion function names, reference samplers.py:47-67) / diffusers scheduler classes (:24-45)
_K_NAMES = {"sample_lms": "lms", "sample_euler": "euler", "sample_euler_ancestral": "euler_a", "sample_dpm_2": "dpm_2",
            "sample_dpm_2_ancestral": "dpm_2_a", "sample_heun": "heun", "sample_dpm_fast": "dpm_fast",
            "sample_dpm_adaptive": "dpm_adaptive", "sample_dpmpp_2s_ancestral": "dpmpp_2s_a", "sample_dpmpp_sde": "dpmpp_sde",
            "sample_dpmpp_2m": "dpmpp_2m"}
_D_NAMES = {"DDIMScheduler": "ddim", "PNDMScheduler": "plms", "LMSDiscreteScheduler": "lms", "EulerDiscreteScheduler": "euler",
            "EulerAncestralDiscreteScheduler": "euler_a", "DPM2DiscreteScheduler": "dpm_2",
            "DPM2AncestralDiscreteScheduler": "dpm_2_a", "HeunDiscreteScheduler": "heun"}


def sampler_name(scheduler: Any) -> str:
    """The native sampler for what the reference's wrapper injected as ``pipeline.scheduler``."""
    if isinstance(scheduler, str):
        return scheduler
    fn = scheduler
    while isinstance(fn, functools.partial):
        fn = fn.func
    name = getattr(fn, "__name__", None)
    if name in _K_NAMES:
        return _K_NAMES[name]
    cls = type(scheduler).__name__
    if cls == "DPMSolverMultistepScheduler":
        cfg = getattr(scheduler, "config", None)
        order = cfg.get("solver_order", 2) if isinstance(cfg, dict) else getattr(cfg, "solver_order", 2)
        return f"dpmsolverpp_{int(order) if isinstance(order, int) else 2}"
    if cls in _D_NAMES:
        return _D_NAMES[cls]
    raise NotImplementedError(f"Scheduler not implemented: {scheduler!r}")


class GyreUnifiedPipeline:
    # reference UnifiedPipeline._meta (unified_pipeline.py:1265): both sampler families; weights stay resident (no offload)
    _meta = {"diffusers_capable": True, "kdiffusion_capable": True, "offload_capable": False}

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler=None, safety_checker=None, feature_extractor=None,
                 clip_model=None, clip_tokenizer=None, inpaint_unet=None, inpaint_text_encoder=None, depth_unet=None,
                 depth_text_encoder=None, hintset_manager=None, text_encoder_2=None, tokenizer_2=None,
                 force_zeros_for_empty_prompt: bool = True, **_unused):
        self.vae, self.text_encoder, self.tokenizer, self.unet = vae, text_encoder, tokenizer, unet
        # SDXL (BASELINE configs[3]; an extension - the reference has no SDXL): the second text tower of the published
        # pipeline layout (model_index.json: text_encoder_2 / tokenizer_2).  Requests are routed by the UNet's
        # addition_embed_type == "text_time" (gyre_amd/text.py SDXLTextConditioner, sdxl_time_ids)
        self.text_encoder_2, self.tokenizer_2 = text_encoder_2, tokenizer_2
        self.force_zeros_for_empty_prompt = force_zeros_for_empty_prompt
        self.scheduler, self.inpaint_unet = scheduler, inpaint_unet
        self.safety_checker, self.feature_extractor = safety_checker, feature_extractor
        self.clip_model, self.clip_tokenizer = clip_model, clip_tokenizer
        self.hintset_manager = hintset_manager
        self.progress_bar: Optional[Callable] = None
        self._grafted_inpaint: Any = False
        self._hires_fix, self._hires_threshold_fraction = True, 0.0333
        self._hires_oos_fraction, self._hires_image_oos_fraction = 0.6, 1.0
        self._text_embedding_layer = "final"
        self._tome = 0
        self._shard_devices: list = []       # engine option "shard_devices": fan one request over these device slots
        self._shard_bit_exact = False
        self._executor = None
        self.clip_default_config = CG.ClipGuidanceConfig()

    # ---- what PipelineWrapper / DiffusionPipelineWrapper touch -----------------------------------------------------------
    def pipeline_modules(self):
        for name in ("vae", "text_encoder", "text_encoder_2", "unet", "inpaint_unet"):
            m = getattr(self, name, None)
            if isinstance(m, torch.nn.Module):
                yield name, m

    @property
    def execution_device(self) -> torch.device:
        return next(self.unet.parameters()).device

    device = execution_device

    def to(self, device):
        for name, m in list(self.pipeline_modules()):
            setattr(self, name, m.to(device))
        return self

    def enable_attention_slicing(self, *_a, **_k): pass
    def disable_attention_slicing(self, *_a, **_k): pass
    def enable_vae_slicing(self): pass
    def disable_vae_slicing(self): pass
    def enable_xformers_memory_efficient_attention(self, *_a, **_k): pass
    def disable_xformers_memory_efficient_attention(self): pass

    def set_options(self, options: dict) -> None:
        """Engine ``options:`` of engines.yaml (reference unified_pipeline.py:1538-1629)."""
        for key, value in options.items():
            if key == "hires":
                if isinstance(value, bool):
                    self._hires_fix = value
                else:
                    for sk, sv in value.items():
                        if sk == "enable":
                            self._hires_fix = bool(sv)
                        elif sk == "oos_fraction":
                            self._hires_oos_fraction = float(sv)
                        elif sk == "image_oos_fraction":
                            self._hires_image_oos_fraction = float(sv)
                        elif sk == "threshold_fraction":
                            self._hires_threshold_fraction = float(sv)
                        else:
                            raise ValueError(f"Unknown option {sk}: {sv} passed as part of hires settings")
            elif key == "grafted_inpaint":
                self._grafted_inpaint = value if isinstance(value, dict) else bool(value)
            elif key == "tome":
                self._tome = int(value) if value else 0
            elif key == "clip_skip":
                self._text_embedding_layer = "penultimate" if value else "final"
            elif key in ("xformers", "vae_tiling", "structured_diffusion"):
                pass                                       # memory knobs / deprecated: accepted, nothing to do
            elif key == "clip":                            # unified_pipeline.py:1591-1623
                cfg = self.clip_default_config
                for sk, sv in (value or {}).items():
                    if sk in ("unet_grad", "vae_grad"):
                        pass                               # native weights never receive gradients; only the sample does
                    elif sk in ("vae_cutouts", "approx_cutouts", "gradient_length"):
                        setattr(cfg, sk, int(sv))
                    elif sk == "no_cutouts":
                        cfg.no_cutouts = bool(sv)
                    elif sk in ("guidance_scale", "gradient_threshold", "gradient_maxloss"):
                        setattr(cfg, sk, float(sv))
                    elif sk == "guidance_base":
                        if str(sv) not in ("guided", "mixed"):
                            raise ValueError("Guidance base must be one of 'mixed' or 'guided'")
                        cfg.guidance_base = str(sv)
                    else:
                        raise ValueError(f"Unknown option {sk}: {sv} passed as part of clip settings")
            elif key == "shard_devices":
                # Extension (north_star: "request batches shard data-parallel across the 8 GPUs of one node"): the images of
                # ONE request are split over these devices with the reference's batched_seeds rule and run concurrently from
                # one host thread + HIP stream per device inside this process (gyre_amd/executor.py); weights are replicated
                # on first use.  {"devices": [...], "bit_exact": true} additionally plans split-K batch-invariantly so that
                # the result equals the single-device run bit for bit.
                devs = value.get("devices", []) if isinstance(value, dict) else (value or [])
                self._shard_bit_exact = bool(value.get("bit_exact", False)) if isinstance(value, dict) else False
                self._shard_devices = [torch.device(d if not isinstance(d, int) else f"cuda:{d}") for d in devs]
                self._executor = None
            elif key == "grafted_depth":
                if value:
                    raise NotImplementedError(f"option {key!r} is outside the native hot path")
            else:
                raise ValueError(f"Unknown option {key}: {value} passed to UnifiedPipeline")

    # ---- text -------------------------------------------------------------------------------------------------------------
    def _embed(self, prompt, negative_prompt, B, num_images_per_prompt, do_cfg, max_embeddings_multiples):
        def frags(p):                                       # str | Prompt | list | PromptBatch -> list of [(text, weight)] / str
            if p is None:
                return None
            if hasattr(p, "as_tokens") and hasattr(p, "prompts"):
                return p.as_tokens()
            if hasattr(p, "as_tokens"):
                return [p.as_tokens()]
            if isinstance(p, (list, tuple)):
                return [q.as_tokens() if hasattr(q, "as_tokens") else q for q in p]
            return [p]
        pos = frags(prompt)
        neg = frags(negative_prompt) or [""] * len(pos)
        if len(neg) == 1 and len(pos) > 1:
            neg = neg * len(pos)
        te, dev = self.text_encoder, self.execution_device
        layer = self._text_embedding_layer

        def encode(ids):
            final = layer == "final"
            out = te(input_ids=ids.to(dev), output_hidden_states=not final, return_dict=True)
            if final:
                return out.last_hidden_state.float()
            tower = getattr(te, "text_model", te)
            return tower.final_layer_norm(out.hidden_states[-2]).float()
        encode.device = dev
        tok = self.tokenizer

        def tokenize(text):
            return tok(text, add_special_tokens=False)["input_ids"] if callable(tok) else tok.encode(text)[1:-1]
        emb = LPWTextEmbedder(encode, tokenize, max_embeddings_multiples)
        cond, unc = emb.get_embeddings(pos, neg if do_cfg else None)
        rep = lambda t: None if t is None else t.repeat_interleave(num_images_per_prompt, dim=0)
        return rep(cond), rep(unc)

    def _embed_sdxl(self, prompt, negative_prompt, B, num_images_per_prompt, do_cfg, max_embeddings_multiples, height, width):
        """SDXL conditioning: context of both text towers [B,77k,2048], pooled text_embeds [B,1280] and time_ids [B,6]
        (gyre_amd/text.py SDXLTextConditioner; published SDXL-base scheme, not in the reference)."""
        from .text import SDXLTextConditioner, sdxl_time_ids
        if self.text_encoder_2 is None or self.tokenizer_2 is None:
            raise ValueError("an SDXL UNet (addition_embed_type 'text_time') needs text_encoder_2 and tokenizer_2")

        def frags(p):
            if p is None:
                return None
            if hasattr(p, "as_tokens") and hasattr(p, "prompts"):
                return p.as_tokens()
            if hasattr(p, "as_tokens"):
                return [p.as_tokens()]
            return list(p) if isinstance(p, (list, tuple)) else [p]
        pos = frags(prompt)
        neg = frags(negative_prompt)        # None stays None: "no negative prompt" conditions on zeros, an explicit "" is encoded

        def mk(tok):
            return lambda text: tok(text, add_special_tokens=False)["input_ids"] if callable(tok) else tok.encode(text)[1:-1]
        dev = self.execution_device
        pad2 = getattr(self.tokenizer_2, "pad_token_id", 0)
        cnd = SDXLTextConditioner(self.text_encoder, mk(self.tokenizer), self.text_encoder_2, mk(self.tokenizer_2), dev,
                                  max_embeddings_multiples, self.force_zeros_for_empty_prompt,
                                  pad_1=getattr(self.tokenizer, "pad_token_id", None), pad_2=0 if pad2 is None else pad2)
        cond, pooled, unc, upooled = cnd(pos, neg, do_cfg)
        rep = lambda t: None if t is None else t.repeat_interleave(num_images_per_prompt, dim=0)
        cond, pooled, unc, upooled = rep(cond), rep(pooled), rep(unc), rep(upooled)
        ids = sdxl_time_ids(cond.shape[0], height, width, device=dev)
        added = {"text_embeds": pooled, "time_ids": ids}
        uadded = {"text_embeds": upooled, "time_ids": ids} if do_cfg else None
        return cond, unc, added, uadded

    # ---- CLIP guidance request (unified_pipeline.py:1876-1909, 1927-1940, 2373-2395) ----------------------------------------
    def _clip_request(self, prompt, clip_prompt, B, num_images_per_prompt, scale, base, glen, gthr, gmax, vae_cutouts,
                      approx_cutouts, no_cutouts) -> dict:
        if scale is not None and self.clip_model is None:
            print("Warning: CLIP guidance passed to a pipeline without a CLIP model. It will be ignored.")
            scale = None
        if not scale:
            return {}
        import copy
        cfg = copy.copy(self.clip_default_config)
        cfg.guidance_scale = scale
        if base is not None:
            cfg.guidance_base = base
        if glen is not None:
            cfg.gradient_length = glen
        if gthr is not None:
            cfg.gradient_threshold = gthr
        if gmax is not None:
            cfg.gradient_maxloss = gmax
        if vae_cutouts is not None:
            cfg.vae_cutouts = vae_cutouts
        if approx_cutouts is not None:
            cfg.guidance_scale = approx_cutouts      # sic: the reference assigns approx_cutouts to guidance_scale (:1899-1900)
        if no_cutouts is not None:
            cfg.no_cutouts = no_cutouts

        def texts(p):
            if hasattr(p, "as_unweighted_string"):
                t = p.as_unweighted_string()
                return t if isinstance(t, list) else [t]
            items = p if isinstance(p, (list, tuple)) else [p]
            return [q.as_unweighted_string() if hasattr(q, "as_unweighted_string") else
                    (q if isinstance(q, str) else " ".join(f[0] for f in q)) for q in items]
        strings = texts(clip_prompt if clip_prompt is not None else prompt)
        if len(strings) * num_images_per_prompt != B:
            raise ValueError(f"clip_prompt has batch size {len(strings)}, but prompt has batch size {B // num_images_per_prompt}. "
                             "They need to match.")
        tok = self.clip_tokenizer if self.clip_tokenizer is not None else self.tokenizer
        te_cfg = getattr(self.text_encoder, "config", None)
        max_pos = getattr(te_cfg, "max_position_embeddings", 77)
        if hasattr(tok, "model_max_length"):
            ids = tok(strings, padding="max_length", max_length=min(tok.model_max_length, max_pos), truncation=True,
                      return_tensors="pt").input_ids
        else:                                       # bare callable tokenizer (tests): BOS + tokens + EOS, padded with EOS
            bos, eos = getattr(te_cfg, "bos_token_id", 49406), getattr(te_cfg, "eos_token_id", 49407)
            rows = []
            for s_ in strings:
                body = list(tok(s_, add_special_tokens=False)["input_ids"])[:max_pos - 2]
                rows.append([bos] + body + [eos] * (max_pos - 1 - len(body)))
            ids = torch.tensor(rows, dtype=torch.long)
        feats = CG._features(self.clip_model.get_text_features(ids.to(self.execution_device)))
        return dict(clip_guidance_scale=scale, clip_config=cfg,
                    clip_text_embeddings=feats.repeat_interleave(num_images_per_prompt, dim=0))

    # ---- the generation call (keywords of reference UnifiedPipeline.__call__, unified_pipeline.py:1722-1790) ---------------
    @torch.no_grad()
    def __call__(self, prompt, height: int = 512, width: int = 512, image=None, mask_image=None, outmask_image=None,
                 depth_map=None, hint_images=None, strength: Optional[float] = None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: int = 1,
                 prediction_type: Optional[str] = "epsilon", eta: Optional[float] = None, churn: Optional[float] = None,
                 churn_tmin: Optional[float] = None, churn_tmax: Optional[float] = None, sigma_min: Optional[float] = None,
                 sigma_max: Optional[float] = None, karras_rho: Optional[float] = None, scheduler_noise_type: str = "normal",
                 generator=None, latents=None, max_embeddings_multiples: int = 3, output_type: str = "pil",
                 return_dict: bool = True, callback=None, callback_steps: int = 1, clip_guidance_scale: Optional[float] = None,
                 clip_guidance_base: Optional[str] = None, clip_gradient_length: Optional[int] = None,
                 clip_gradient_threshold: Optional[float] = None, clip_gradient_maxloss: Optional[float] = None,
                 clip_prompt=None, vae_cutouts: Optional[int] = None, approx_cutouts: Optional[int] = None, no_cutouts=False,
                 run_safety_checker: bool = True, lora=None, token_embeddings=None,
                 hires_fix=None, hires_oos_fraction=None, tiling=False, debug_latent_tags=None, debug_latent_prefix="",
                 cfg_execution: str = "parallel"):
        if depth_map is not None or hint_images:
            raise NotImplementedError("depth / hint conditioning (ControlNet, T2I adapters) is outside the native hot path")
        if token_embeddings:
            raise NotImplementedError("textual-inversion token embeddings are outside the native hot path")
        if tiling not in (False, None, True, "x", "y", "xy"):
            raise ValueError(f"tiling must be True, False, 'x', 'y' or 'xy', got {tiling!r}")
        if tiling and clip_guidance_scale:
            raise NotImplementedError("tiling together with CLIP guidance (the native input-gradient sweep has no circular convolutions)")
        if scheduler_noise_type not in (None, "normal", "brownian"):
            raise ValueError(f"scheduler_noise_type must be 'normal' or 'brownian', got {scheduler_noise_type!r}")
        # `latents`: accepted and IGNORED, as in the reference - UnifiedPipeline.__call__ declares and documents the keyword
        # (unified_pipeline.py:1749,1807-1810) but never reads it; start latents always come from the per-image generators.
        # A caller that passes start latents gets an image unrelated to them, so say it once per call (parity kept, silence not)
        if latents is not None:
            import warnings
            warnings.warn("GyreUnifiedPipeline: the `latents` argument is accepted for signature compatibility and ignored (as in the "
                          "reference, unified_pipeline.py:1749); start latents come from the per-image generators / seeds",
                          RuntimeWarning, stacklevel=2)
        if self.scheduler is None:
            raise ValueError("no scheduler injected")
        n_prompts = len(prompt.prompts) if hasattr(prompt, "prompts") else (len(prompt) if isinstance(prompt, (list, tuple)) else 1)
        B = n_prompts * num_images_per_prompt
        if generator is None:
            generators = [torch.Generator("cpu").manual_seed(torch.seed() % (2 ** 31)) for _ in range(B)]
        else:
            generators = list(generator) if isinstance(generator, (list, tuple)) else [generator]
            if len(generators) != B:
                if len(generators) == 1 and B > 1:
                    raise ValueError("one torch.Generator per image is required for batch-independent results")
                raise ValueError(f"Generator passed as a list, but list length does not match batch size {B}")
        do_cfg = guidance_scale > 1.0
        sdxl = getattr(getattr(self.unet, "config", None), "addition_embed_type", None) == "text_time"
        added = uadded = None
        if sdxl:
            cond, unc, added, uadded = self._embed_sdxl(prompt, negative_prompt, B, num_images_per_prompt, do_cfg,
                                                        max_embeddings_multiples, height, width)
        else:
            cond, unc = self._embed(prompt, negative_prompt, B, num_images_per_prompt, do_cfg, max_embeddings_multiples)
        if strength is None:
            strength = 0.8
        dev = self.execution_device
        pipe = GyrePipeline(self.unet, self.vae, None, device=dev, inpaint_unet=self.inpaint_unet,
                            grafted_inpaint=self._grafted_inpaint, clip_model=self.clip_model,
                            feature_extractor=self.feature_extractor)
        clip_kw = self._clip_request(prompt, clip_prompt, B, num_images_per_prompt, clip_guidance_scale, clip_guidance_base,
                                     clip_gradient_length, clip_gradient_threshold, clip_gradient_maxloss, vae_cutouts,
                                     approx_cutouts, no_cutouts)
        pipe.hires_fix, pipe.hires_threshold_fraction = self._hires_fix, self._hires_threshold_fraction
        pipe.hires_oos_fraction, pipe.hires_image_oos_fraction = self._hires_oos_fraction, self._hires_image_oos_fraction
        # reference set_tiling_mode (unified_pipeline.py:1696-1712, called per request at :1845): every Conv2d of every module
        # pads circularly for this request - the native modules switch their conv gather
        for m in (self.unet, self.inpaint_unet, self.vae):
            if m is not None and hasattr(m, "set_tiling"):
                m.set_tiling(tiling or False)
        for u in (self.unet, self.inpaint_unet):
            if u is None:
                continue
            LR.remove_lora_from_model(u)                    # the reference strips leftovers on every call (:2190-2200)
            if hasattr(u, "set_tome"):                      # the reference patches ToMe into self.unet only (:1580-1584)
                u.set_tome(self._tome if u is self.unet else 0)
        if lora:
            for i, spec in enumerate(lora if isinstance(lora, (list, tuple)) else [lora]):
                tensors, weights = (spec if isinstance(spec, (list, tuple)) else (spec, {}))
                scale = (weights or {}).get("unet", 1.0) if isinstance(weights, dict) else 1.0
                LR.apply_lora(self.unet, tensors, f"request-{i}", scale)
        steps_seen = []

        def cb(info):
            steps_seen.append(info.get("i", len(steps_seen)))
            if callback is not None and len(steps_seen) % max(callback_steps, 1) == 0:
                callback(len(steps_seen) - 1, info.get("t"), info.get("x"))
        if self.progress_bar is not None:
            # the wrapper's tqdm object polls the stop event on every update (ProgressBarWrapper, pipeline_wrapper.py:26-47)
            bar = self.progress_bar(total=num_inference_steps)
            cb_inner = cb

            def cb(info, _b=bar):
                cb_inner(info)
                if hasattr(_b, "update"):
                    _b.update(1)
        to_dev = lambda t: None if t is None else t.to(dev)
        request = dict(generators=generators, text_embeddings=cond, uncond_embeddings=unc, height=height, width=width,
                       num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                       sampler=sampler_name(self.scheduler), image=to_dev(image), mask_image=to_dev(mask_image),
                       strength=strength, karras_rho=karras_rho, eta=eta, cfg_execution=cfg_execution,
                       hires_fix=hires_fix, hires_oos_fraction=hires_oos_fraction,
                       prediction_type=prediction_type or "epsilon", churn=churn, churn_tmin=churn_tmin or 0.0,
                       churn_tmax=churn_tmax if churn_tmax is not None else float("inf"), sigma_min=sigma_min,
                       sigma_max=sigma_max, scheduler_noise_type=scheduler_noise_type or "normal", **clip_kw)
        if sdxl:
            request.update(added_cond=added, uncond_added_cond=uadded)
        if len(self._shard_devices) > 1 and B > 1 and not clip_kw and not lora and not self._tome and not tiling:
            # one request over several device slots (engine option "shard_devices"); progress / cancellation are polled once
            # per request here: the replicas run their loops concurrently and a per-step callback has no single owner.
            # (CLIP guidance couples the batch through its flat-loss stop; per-request LoRA / ToMe patch one
            # UNet object - such requests run on this slot alone)
            images = self._sharded(pipe, request, outmask_image=to_dev(outmask_image), image=to_dev(image))
            cb({"i": num_inference_steps - 1})
        else:
            images = pipe(callback=cb, outmask_image=to_dev(outmask_image), **request)
        images = images.float().cpu()                       # reference: result_image.cpu() ... BCHW 0..1 (:2512-2531)
        images, nsfw = self._safety_check(images, run_safety_checker)
        if output_type == "pil":
            images = self.numpy_to_pil(images.permute(0, 2, 3, 1).numpy())
        elif output_type not in ("tensor", "pt"):
            raise ValueError(f"output_type {output_type!r}")
        if not return_dict:
            return images, nsfw
        from types import SimpleNamespace
        return SimpleNamespace(images=images, nsfw_content_detected=nsfw)

    def _sharded(self, pipe, request, outmask_image=None, image=None):
        from . import images as I
        from .executor import DeviceSlotExecutor
        ex = self._executor
        if ex is not None and (ex.source.unet is not self.unet or ex.source.vae is not self.vae
                               or ex.source.inpaint_unet is not self.inpaint_unet or ex.stale(self._shard_devices)):
            ex = None       # other module objects, new weights (load_state_dict / .to / .half / LoRA) or another device list
        if ex is None:
            pipe0 = GyrePipeline(self.unet, self.vae, None, device=self.execution_device, inpaint_unet=self.inpaint_unet,
                                 grafted_inpaint=self._grafted_inpaint)
            self._executor = DeviceSlotExecutor.replicate(pipe0, self._shard_devices)
        ex = self._executor
        for rep in ex.pipelines:                 # per-request engine settings follow the replicas
            rep.grafted_inpaint = self._grafted_inpaint
            rep.hires_fix, rep.hires_threshold_fraction = pipe.hires_fix, pipe.hires_threshold_fraction
            rep.hires_oos_fraction, rep.hires_image_oos_fraction = pipe.hires_oos_fraction, pipe.hires_image_oos_fraction
        latents = ex(bit_exact=self._shard_bit_exact, **request)
        result = ex.decode(latents)
        if image is not None and outmask_image is not None:         # unified_pipeline.py:2493-2510
            src = image if image.ndim == 4 else image[None]
            om = outmask_image if outmask_image.ndim == 4 else outmask_image[None]
            result = I.outmask_composite(result, src.to(result.device), om.to(result.device))
        return result

    @staticmethod
    def numpy_to_pil(images):
        """NHWC 0..1 floats -> PIL images (what the reference inherits from DiffusionPipeline.numpy_to_pil)."""
        from PIL import Image
        if images.ndim == 3:
            images = images[None]
        return [Image.fromarray(im) for im in (images * 255).round().astype("uint8")]

    def _safety_check(self, images: torch.Tensor, run_safety_checker: bool):
        """The host-side NSFW check of the reference (unified_pipeline.py:2512-2523): whenever a checker is loaded it sees
        the finished images - ``feature_extractor`` on the PIL images, then ``safety_checker(images=<NHWC numpy>,
        clip_input=<pixel_values in the text-embedding dtype>)`` - and its (possibly blacked-out) images and per-image flags
        are what the call returns.  The server's default ``nsfw_behaviour`` is "block" (server.py:660) and relies on it."""
        if not run_safety_checker or self.safety_checker is None:
            return images, [False] * images.shape[0]
        if self.feature_extractor is None:
            raise ValueError("a safety_checker needs the pipeline's feature_extractor")
        result_numpy = images.permute(0, 2, 3, 1).numpy()
        dev = self.execution_device
        te_param = next(iter(self.text_encoder.parameters()), None) if hasattr(self.text_encoder, "parameters") else None
        dtype = te_param.dtype if te_param is not None else torch.float32
        clip_input = self.feature_extractor(self.numpy_to_pil(result_numpy), return_tensors="pt").to(dev)
        result_numpy, nsfw = self.safety_checker(images=result_numpy, clip_input=clip_input.pixel_values.to(dtype))
        if isinstance(result_numpy, torch.Tensor):
            result_numpy = result_numpy.float().cpu().numpy()
        import numpy as np
        return torch.from_numpy(np.ascontiguousarray(result_numpy)).permute(0, 3, 1, 2), [bool(f) for f in nsfw]
