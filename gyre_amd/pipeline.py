"""Diffusers-free generation pipeline over the native UNet / VAE.

Host orchestration restating the reference's ``UnifiedPipeline.__call__``
(gyre/pipeline/unified_pipeline.py:1722-2538) for the modes on the hot path:

  Txt2imgMode                 unified_pipeline.py:155-237
  Img2imgMode                 unified_pipeline.py:240-337
  EnhancedInpaintMode         unified_pipeline.py:398-645 (per-step blend of original vs predicted latents)
  EnhancedRunwayInpaintMode   unified_pipeline.py:648-696 (9-channel UNet input assembly)
  strength >= 1 "shaped noise" fill (noise_mode 5)    unified_pipeline.py:402-417, 466-607
  Hires fix (mode tree of a natural-size and a full-size leaf)   unified_pipeline.py:1064-1200, 2100-2181
  Grafted inpaint (inpaint_unet + unet blended by GraftUnets)    unified_pipeline.py:2071-2100, unet/graft.py:16-56
  CLIP guidance (ClipGuidedMode around every leaf)               unified_pipeline.py:2373-2406, unet/clipguided.py

Call stack per SURVEY.md 3.2/3.3: embeddings -> UNetWithEmbeddings -> (UnetWithExtraChannels)
-> CFGUNet_Parallel -> KDiffusionUNetWrapper -> sampler loop -> vae.decode(latents / 0.18215)
-> (x/2+0.5).clamp(0,1).  Everything except the UNet / VAE forward is thin PyTorch host code.

Random draws are per image (one generator per seed, reference pipeline_wrapper.py:243-253 +
randtools.py:39-64), which is what makes data-parallel sharding exact: an image's result
does not depend on which other images share its batch or GPU.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Callable, List, Optional, Sequence

import torch

from . import clipguided as CG
from . import hires as H
from . import images as I
from . import schedulers as S

Tensor = torch.Tensor


def build_generators(seeds: Sequence[int], device="cpu") -> List[torch.Generator]:
    return [torch.Generator(device).manual_seed(int(s)) for s in seeds]


def downscale_boxop_1d(inp: Tensor, scale: int = 8, op: str = "max") -> Tensor:
    shape = inp.shape[:-1] + (inp.shape[-1] // scale, scale)
    return getattr(inp.reshape(shape), op)(dim=-1).values


def downscale_boxop_2d(inp: Tensor, scale: int = 8, op: str = "max") -> Tensor:
    mid = downscale_boxop_1d(inp, scale, op)
    return downscale_boxop_1d(mid.transpose(-2, -1), scale, op).transpose(-2, -1)


def mask_to_latent_mask(mask: Tensor, inputIs1K0D: bool = True) -> Tensor:
    mask = downscale_boxop_2d(mask, 8, "min" if inputIs1K0D else "max")
    return mask[:, [0, 0, 0, 0]]


def round_mask(mask: Tensor, threshold: float = 0.5) -> Tensor:
    mask = mask.clone()
    mask[mask >= threshold] = 1
    mask[mask < 1] = 0
    return mask


def fill_with_shaped_noise(init_latents: Tensor, latent_mask: Tensor, generators, shaped_noise_strength: float) -> Tensor:
    """EnhancedInpaintMode._fillWithShapedNoise, noise_mode 5 (unified_pipeline.py:466-601): the repaint area of the
    clean init latents is filled, per image and channel, with pixels drawn at random from the protected area (numpy
    Generator seeded from the image's torch generator), mixed with white noise by ``shaped_noise_strength``."""
    import numpy as np
    high = round_mask(latent_mask, 0.001)
    good = high[0, 0].ge(0.5)                                 # first image's mask selects the donor pixels
    if not bool(good.any()):
        raise ValueError("shaped-noise fill needs at least one protected latent cell in the mask")
    s = float(shaped_noise_strength)
    rows = []
    for g, lat in zip(generators, (init_latents * high).split(1)):
        seed = torch.randint(low=0, high=torch.iinfo(torch.int32).max, size=[1], generator=g, device=g.device,
                             dtype=torch.int32).cpu()
        npgen = np.random.default_rng(seed.numpy())
        chans = [torch.from_numpy(npgen.choice(ch[0, 0][good].cpu().numpy(), tuple(ch.shape))).to(lat.device, lat.dtype)
                 for ch in lat.split(1, dim=1)]
        white = torch.zeros(lat.shape, dtype=lat.dtype, device=g.device).normal_(generator=g).to(lat.device)
        rows.append(white * (1 - s) + torch.cat(chans, dim=1) * s)
    noise = torch.cat(rows, dim=0)
    return init_latents * latent_mask + noise * (1 - latent_mask)


def txt2img_latents(generators, channels: int, lat_h: int, lat_w: int, unet_sample_size: int, device,
                    dtype=torch.float32) -> Tensor:
    """Always draw the sample_size^2 noise first, then crop / embed (unified_pipeline.py:193-234)."""
    B = len(generators)
    mid = S.batched_randn([B, channels, unet_sample_size, unet_sample_size], generators, device, dtype)
    off2 = (unet_sample_size - lat_h) // 2
    off3 = (unet_sample_size - lat_w) // 2
    if off2 > 0:
        mid = mid[:, :, off2:off2 + lat_h, :]
    if off3 > 0:
        mid = mid[:, :, :, off3:off3 + lat_w]
    if off2 >= 0 and off3 >= 0:
        return mid.contiguous()
    latents = S.batched_randn([B, channels, lat_h, lat_w], generators, device, dtype)
    o2 = (latents.shape[2] - mid.shape[2]) // 2
    o3 = (latents.shape[3] - mid.shape[3]) // 2
    latents[:, :, o2:o2 + mid.shape[2], o3:o3 + mid.shape[3]] = mid
    return latents


class GyrePipeline:
    """unet / vae: GyreHipUNet / GyreHipVAE (or anything with the same call contract, e.g. the
    oracle adapters used by the tests).  text_encoder: optional callable ids[B,77] -> [B,77,D]."""

    vae_scale_factor = 8
    latent_scale = 0.18215          # overridden per instance from vae.config.scaling_factor
    # hires-fix engine defaults (reference unified_pipeline.py:1368-1373)
    hires_fix = True
    hires_threshold_fraction = 0.0333
    hires_oos_fraction = 0.6
    hires_image_oos_fraction = 1.0

    def __init__(self, unet, vae, text_encoder: Optional[Callable] = None, device="cuda:0", inpaint_unet=None,
                 grafted_inpaint=False, clip_model=None, feature_extractor=None):
        """inpaint_unet: optional 9-channel (runway) UNet used whenever a mask is given (reference
        unified_pipeline.py:1352,2058-2062).  grafted_inpaint: True or a blend dict {floor, start, end, easing}: masked
        requests then run BOTH UNets and GraftUnets hands over from the inpaint UNet to `unet` (reference option
        "grafted_inpaint", unified_pipeline.py:1543, 2082-2098).
        clip_model / feature_extractor: the CLIP model (``get_image_features`` / ``get_text_features``, host PyTorch)
        and its preprocessing constants (``image_mean``, ``image_std``, ``size``) for CLIP guidance (reference
        unified_pipeline.py:1347-1349, 1407-1411); without them ``clip_guidance_scale`` is ignored with a warning, as the
        reference does (:1877-1881)."""
        self.unet, self.vae, self.text_encoder = unet, vae, text_encoder
        self.clip_model, self.feature_extractor = clip_model, feature_extractor
        self.clip_default_config = CG.ClipGuidanceConfig()
        self.inpaint_unet, self.grafted_inpaint = inpaint_unet, grafted_inpaint
        self.device = torch.device(device)
        # the reference hard-codes SD1.x's 0.18215 (unified_pipeline.py:319,2488); SDXL's VAE uses 0.13025
        self.latent_scale = float(getattr(getattr(vae, "config", None), "scaling_factor", 0.18215))

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype: Optional[torch.dtype] = torch.bfloat16, device="cuda:0",
                        variant: Optional[str] = None, text_encoder: Optional[Callable] = None,
                        inpaint_unet: Optional[str] = None, clip_model: Optional[str] = None, grafted_inpaint=False):
        """Build the pipeline from a diffusers-layout model folder (model_index.json, unet/, vae/[, text_encoder/])
        without importing diffusers - the layout the reference's manager resolves engines to (manager.py:1024-1252).
        The CLIP text encoder is loaded through transformers when its folder is present and no callable is given.
        inpaint_unet / clip_model: folders of the engine's optional overrides (reference tests/engines.clip.yaml:12-17 -
        a 9-channel inpaint UNet, a transformers CLIPModel + its preprocessor config for CLIP guidance)."""
        import os
        from .modules import GyreHipUNet, GyreHipVAE
        if not os.path.exists(os.path.join(path, "model_index.json")) and not os.path.isdir(os.path.join(path, "unet")):
            raise FileNotFoundError(f"{path} is not a diffusers model folder (no model_index.json / unet/)")
        unet = GyreHipUNet.from_pretrained(path, subfolder="unet", torch_dtype=torch_dtype, variant=variant)
        vae = GyreHipVAE.from_pretrained(path, subfolder="vae", torch_dtype=torch_dtype, variant=variant)
        dev = torch.device(device)
        if dev.type == "cuda":
            unet, vae = unet.to(dev), vae.to(dev)
        te_dir = os.path.join(path, "text_encoder")
        if text_encoder is None and os.path.isdir(te_dir):
            from transformers import CLIPTextModel
            te = CLIPTextModel.from_pretrained(te_dir, torch_dtype=torch_dtype).to(dev).eval()
            text_encoder = lambda ids: te(input_ids=ids)[0]
        iu = None
        if inpaint_unet:
            iu = GyreHipUNet.from_pretrained(inpaint_unet, torch_dtype=torch_dtype, variant=variant)
            iu = iu.to(dev) if dev.type == "cuda" else iu
        cm = fe = None
        if clip_model:
            import json
            from types import SimpleNamespace
            from transformers import CLIPModel
            from .clipguided import patch_embedding_as_matmul
            cm = patch_embedding_as_matmul(CLIPModel.from_pretrained(clip_model).to(dev).eval().requires_grad_(False))
            pj = os.path.join(clip_model, "preprocessor_config.json")
            pc = json.load(open(pj)) if os.path.exists(pj) else {}
            fe = SimpleNamespace(image_mean=pc.get("image_mean", [0.48145466, 0.4578275, 0.40821073]),
                                 image_std=pc.get("image_std", [0.26862954, 0.26130258, 0.27577711]),
                                 size=pc.get("size", {"shortest_edge": 224}))
        return cls(unet, vae, text_encoder, device=dev, inpaint_unet=iu, grafted_inpaint=grafted_inpaint, clip_model=cm,
                   feature_extractor=fe)

    # -- text ---------------------------------------------------------------------------------
    def encode_ids(self, input_ids: Tensor) -> Tensor:
        if self.text_encoder is None:
            raise ValueError("no text_encoder configured: pass text_embeddings instead of input_ids")
        return self.text_encoder(input_ids.to(self.device))

    # -- latents <-> image --------------------------------------------------------------------------
    def vae_decode(self, latents: Tensor) -> Tensor:
        img = self.vae.decode(latents / self.latent_scale).sample  # unified_pipeline.py:1523-1536, :2488
        return (img / 2 + 0.5).clamp(0, 1)

    def image_to_latents(self, image: Tensor, generators, mask: Optional[Tensor] = None) -> Tensor:
        """Img2imgMode._convertToLatents: one posterior sample per image generator, x 0.18215."""
        image = image.to(self.device, torch.float32)
        if mask is not None:
            image = image * (mask.to(self.device) > 0.5)
        dist = self.vae.encode(image).latent_dist
        latents = torch.cat([dist.sample(generator=g) for g in generators], dim=0)
        return self.latent_scale * latents.to(self.device, torch.float32)

    @staticmethod
    def preprocess_image(t: Tensor) -> Tensor:
        if t.ndim == 3:
            t = t[None]
        return 2.0 * t[:, [0, 1, 2]] - 1.0

    @staticmethod
    def preprocess_mask(t: Tensor, inputIs0K1D: bool = True) -> Tensor:
        if t.ndim == 3:
            t = t[None]
        t = t[:, [0]]
        return 1 - t if inputIs0K1D else t

    # -- mode tree -----------------------------------------------------------------------------------
    # reference ModeTreeRoot / Node / Leaf (unified_pipeline.py:1064-1200): leaves are modes (one UNet stack each), inner
    # nodes blend two sub-trees during sampling (GraftUnets, HiresUnetWrapper).  A tree here is a _Leaf or a
    # (left, right, merger class, merger kwargs) tuple.
    class _Leaf(SimpleNamespace):
        pass

    @staticmethod
    def _leaves(tree):
        if isinstance(tree, tuple):
            return GyrePipeline._leaves(tree[0]) + GyrePipeline._leaves(tree[1])
        return [tree]

    @staticmethod
    def _map_leaves(tree, fn):
        if isinstance(tree, tuple):
            return (GyrePipeline._map_leaves(tree[0], fn), GyrePipeline._map_leaves(tree[1], fn), tree[2], tree[3])
        return fn(tree)

    def _construct_leaf(self, leaf, generators, B):
        """Mode constructor (Img2imgMode / EnhancedInpaintMode / EnhancedRunwayInpaintMode __init__): preprocessing and,
        for the inpaint modes, the posterior sample of the MASKED original (init_latents_orig, unified_pipeline.py:432-434).
        Constructors of every leaf run before any leaf generates its start latents (build_mode, :1108-1116), which fixes
        the order of the per-image generator draws."""
        dev = self.device
        leaf.lat_h, leaf.lat_w = leaf.height // self.vae_scale_factor, leaf.width // self.vae_scale_factor
        leaf.extra = leaf.blend_orig = leaf.blend_mask = leaf.noise = leaf.latent_mask = leaf.img = None
        runway = leaf.unet.config.in_channels == 9
        if leaf.image is None:
            if runway:
                raise ValueError("the 9-channel inpaint UNet needs image and mask_image")
            return
        img = self.preprocess_image(leaf.image.to(torch.float32))
        if img.shape[-2:] != (leaf.height, leaf.width):
            raise ValueError(f"image is {tuple(img.shape[-2:])}, expected {(leaf.height, leaf.width)}")
        leaf.img = img
        if leaf.mask_image is None:
            if runway:
                raise ValueError("the 9-channel inpaint UNet needs image and mask_image")
            return
        mask = self.preprocess_mask(leaf.mask_image.to(torch.float32)).to(dev)      # 1 keep / 0 replace
        orig = self.image_to_latents(img, generators, round_mask(mask, 0.001))
        leaf.latent_mask = torch.cat([mask_to_latent_mask(mask)] * B)
        if runway and not leaf.enhanced:
            inpaint_mask = 1 - round_mask(leaf.latent_mask, 0.001)[:, [0]]            # 0 keep / 1 replace
            leaf.extra = torch.cat([inpaint_mask, orig], dim=1)
        else:
            # EnhancedInpaintMode: keep the protected area pinned to the (masked) original by blending the
            # denoised prediction with it while the blend mask exceeds the progress u (_blend, :620-625)
            leaf.blend_orig, leaf.blend_mask = orig, leaf.latent_mask

    def _bind_leaf(self, leaf, *, text_embeddings, uncond_embeddings, guidance_scale, cfg_execution, B, added_cond,
                   uncond_added_cond, cfg_embeddings, clip_mode=None):
        """UNet stack of one leaf: embeddings -> extra channels -> CFG (unified_pipeline.py:2235-2337, 2408-2430)."""
        def bind(emb, added=None):
            u = S.UNetWithEmbeddings(leaf.unet, emb, added)
            return S.UnetWithExtraChannels(u, leaf.extra) if leaf.extra is not None else u

        if guidance_scale > 1.0:
            if cfg_execution == "sequential":
                leaf.eps_unet = S.CFGUNet_Sequential(bind(text_embeddings, added_cond), bind(uncond_embeddings, uncond_added_cond),
                                                     guidance_scale, B)
            else:
                both = None
                if added_cond is not None:
                    both = {k: torch.cat([uncond_added_cond[k], added_cond[k]]) for k in added_cond}
                # ONE concatenated tensor for every leaf: the native UNet recognises the context it already projected by
                # tensor identity, and the leaves of a hires / graft tree alternate on the same UNet every step
                if cfg_embeddings.get("both") is None:
                    cfg_embeddings["both"] = torch.cat([uncond_embeddings, text_embeddings])
                leaf.eps_unet = S.CFGUNet_Parallel(bind(cfg_embeddings["both"], both), guidance_scale, B)
        else:
            leaf.eps_unet = bind(text_embeddings, added_cond)
        leaf.clip_mode = clip_mode
        if clip_mode is not None:       # ClipGuidedMode.wrap_guidance_unet over the conditional / unconditional stems
            cfg = guidance_scale > 1.0
            both_unet = None
            if cfg and cfg_execution != "sequential":      # one activation-keeping pass over cat[uncond, cond] per guided step
                both_added = None
                if added_cond is not None:
                    both_added = {k: torch.cat([uncond_added_cond[k], added_cond[k]]) for k in added_cond}
                both_unet = bind(cfg_embeddings["both"], both_added)
            leaf.eps_unet = clip_mode.wrap_guidance_unet(bind(text_embeddings, added_cond),
                                                         bind(uncond_embeddings, uncond_added_cond) if cfg else None,
                                                         leaf.eps_unet, guidance_scale, unet_both=both_unet)

    def _generate_leaf_latents(self, leaf, sched, generators, fill_strength):
        """Mode.generateLatents: txt2img noise, or (init sample, optional shaped-noise fill, noise) for an init image."""
        dev = self.device
        if leaf.img is None:
            sample_size = getattr(leaf.unet.config, "sample_size", 64)
            latents = txt2img_latents(generators, 4, leaf.lat_h, leaf.lat_w, sample_size, dev)
            return sched.prepare_initial_latents(latents)
        init = self.image_to_latents(leaf.img, generators)
        if leaf.latent_mask is not None and fill_strength is not None:    # strength >= 1 (unified_pipeline.py:603-607)
            init = fill_with_shaped_noise(init, leaf.latent_mask, generators, fill_strength)
        leaf.noise = S.batched_randn(init.shape, generators, dev, torch.float32)
        return sched.add_noise(init, leaf.noise)

    # -- the generation call ------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, *, seeds: Optional[Sequence[int]] = None, generators: Optional[Sequence[torch.Generator]] = None,
                 text_embeddings: Optional[Tensor] = None,
                 uncond_embeddings: Optional[Tensor] = None, input_ids: Optional[Tensor] = None,
                 negative_ids: Optional[Tensor] = None, height: int = 512, width: int = 512,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, sampler: str = "dpmpp_2m",
                 image: Optional[Tensor] = None, mask_image: Optional[Tensor] = None, strength: float = 0.8,
                 karras_rho: Optional[float] = None, eta: Optional[float] = None, cfg_execution: str = "parallel",
                 output_type: str = "image", callback=None, generator_device: str = "cpu",
                 hires_fix: Optional[bool] = None, hires_oos_fraction: Optional[float] = None,
                 outmask_image: Optional[Tensor] = None, added_cond: Optional[dict] = None,
                 uncond_added_cond: Optional[dict] = None, prediction_type: str = "epsilon",
                 churn: Optional[float] = None, churn_tmin: float = 0.0, churn_tmax: float = float("inf"),
                 sigma_min: Optional[float] = None, sigma_max: Optional[float] = None, scheduler_noise_type: str = "normal",
                 clip_guidance_scale: Optional[float] = None, clip_guidance_base: Optional[str] = None,
                 clip_gradient_length: Optional[int] = None, clip_gradient_threshold: Optional[float] = None,
                 clip_gradient_maxloss: Optional[float] = None, vae_cutouts: Optional[int] = None,
                 approx_cutouts: Optional[int] = None, no_cutouts=None, clip_input_ids: Optional[Tensor] = None,
                 clip_text_embeddings: Optional[Tensor] = None, clip_config: Optional[CG.ClipGuidanceConfig] = None):
        """clip_*: CLIP guidance (reference keywords of UnifiedPipeline.__call__, unified_pipeline.py:1756-1764).  The text
        side comes as ``clip_input_ids`` (encoded by clip_model.get_text_features) or ``clip_text_embeddings`` [B, D]."""
        if height % self.vae_scale_factor or width % self.vae_scale_factor:
            raise ValueError(f"`height` and `width` have to be divisible by {self.vae_scale_factor} "
                             f"but are {height} and {width}.")
        if (seeds is None) == (generators is None):
            raise ValueError("pass either seeds or generators (one per image)")
        B = len(seeds) if seeds is not None else len(generators)
        if B < 1:
            raise ValueError("at least one seed (image) is required")
        dev = self.device
        # the reference hands the pipeline one torch.Generator per image (pipeline_wrapper.py:243-253)
        generators = build_generators(seeds, generator_device) if seeds is not None else list(generators)
        if text_embeddings is None:
            if input_ids is None:
                raise ValueError("pass text_embeddings or input_ids")
            text_embeddings = self.encode_ids(input_ids)
        text_embeddings = text_embeddings.to(dev)
        if text_embeddings.shape[0] == 1 and B > 1:
            text_embeddings = text_embeddings.expand(B, -1, -1)
        if text_embeddings.shape[0] != B:
            raise ValueError(f"text_embeddings batch {text_embeddings.shape[0]} != number of seeds {B}")
        do_cfg = guidance_scale > 1.0
        if do_cfg:
            if uncond_embeddings is None:
                if negative_ids is None:
                    raise ValueError("guidance_scale > 1 needs uncond_embeddings or negative_ids")
                uncond_embeddings = self.encode_ids(negative_ids)
            uncond_embeddings = uncond_embeddings.to(dev)
            if uncond_embeddings.shape[0] == 1 and B > 1:
                uncond_embeddings = uncond_embeddings.expand(B, -1, -1)

        # SDXL added conditioning (BASELINE config 4; not in the reference): {"text_embeds": [B or 1, D], "time_ids": [B or 1, 6]}
        if getattr(self.unet.config, "addition_embed_type", None) == "text_time":
            if added_cond is None:
                raise ValueError("this UNet needs added_cond={'text_embeds': ..., 'time_ids': ...}")
            exp = lambda d: {k: (v.to(dev).expand(B, -1) if v.shape[0] == 1 and B > 1 else v.to(dev)) for k, v in d.items()}
            added_cond = exp(added_cond)
            if do_cfg:
                uncond_added_cond = exp(uncond_added_cond if uncond_added_cond is not None else
                                        {"text_embeds": torch.zeros_like(added_cond["text_embeds"]), "time_ids": added_cond["time_ids"]})
            for d in (added_cond, uncond_added_cond if do_cfg else None):
                if d is not None and any(v.shape[0] != B for v in d.values()):
                    raise ValueError("added_cond tensors must have batch 1 or the number of seeds")
        else:
            added_cond = uncond_added_cond = None
        fill_strength = None
        if image is not None:
            if mask_image is not None:
                # inpaint modes accept strength in [0, 2]: from 1 up the repaint area is first refilled with shaped
                # noise, and the share of white noise in that fill grows to 100 % at 2 (unified_pipeline.py:402-417)
                if strength < 0 or strength > 2:
                    raise ValueError(f"The value of strength should in [0.0, 2.0] but is {strength}")
                if strength >= 1.0:
                    fill_strength = min(2 - strength, 1)
                strength = min(strength, 1)
            elif strength < 0 or strength > 1:
                raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")

        sched = S.make_scheduler(sampler, generators, dev, torch.float32)
        is_k = isinstance(sched, S.KDiffusionScheduler)

        # ---- CLIP guidance configuration (unified_pipeline.py:1876-1909, 2373-2395) ----
        if clip_guidance_scale is not None and (self.clip_model is None or self.feature_extractor is None):
            print("Warning: CLIP guidance passed to a pipeline without a CLIP model. It will be ignored.")
            clip_guidance_scale = None
        if not clip_guidance_scale:
            clip_config = None
        if clip_guidance_scale:
            import copy
            if clip_config is None:                  # (an engine adapter hands over its resolved configuration instead)
                clip_config = copy.copy(self.clip_default_config)
                clip_config.guidance_scale = clip_guidance_scale
            for name, val in (("guidance_base", clip_guidance_base), ("gradient_length", clip_gradient_length),
                              ("gradient_threshold", clip_gradient_threshold), ("gradient_maxloss", clip_gradient_maxloss),
                              ("vae_cutouts", vae_cutouts), ("approx_cutouts", approx_cutouts), ("no_cutouts", no_cutouts)):
                if val is not None:
                    setattr(clip_config, name, val)
            if clip_config.guidance_base not in ("guided", "mixed"):
                raise ValueError(f"clip_guidance_base must be 'guided' or 'mixed', got {clip_config.guidance_base!r}")
            if not is_k and prediction_type == "v_prediction":
                raise ValueError("Can't use Diffusers scheduler with a v-prediction unet and CLIP guidance. "
                                 "Either use a K-Diffusion scheduler or don't use CLIP guidance.")
            if clip_text_embeddings is None:
                if clip_input_ids is None:
                    raise ValueError("CLIP guidance needs clip_input_ids or clip_text_embeddings")
                clip_text_embeddings = CG._features(self.clip_model.get_text_features(clip_input_ids.to(dev)))
            clip_text_embeddings = clip_text_embeddings.to(dev, torch.float32)
            clip_text_embeddings = clip_text_embeddings / clip_text_embeddings.norm(p=2, dim=-1, keepdim=True)
            if clip_text_embeddings.shape[0] == 1 and B > 1:
                clip_text_embeddings = clip_text_embeddings.expand(B, -1)
            if clip_text_embeddings.shape[0] != B:
                raise ValueError(f"clip prompt batch {clip_text_embeddings.shape[0]} != number of seeds {B}")

        # ---- mode tree (unified_pipeline.py:2054-2181) --------------------------------------------------------
        main_unet = self.inpaint_unet if (mask_image is not None and self.inpaint_unet is not None) else self.unet
        tree = self._Leaf(unet=main_unet, enhanced=False, height=height, width=width, image=image, mask_image=mask_image)
        if main_unet is self.inpaint_unet and main_unet is not self.unet and self.grafted_inpaint:
            blend = self.grafted_inpaint if isinstance(self.grafted_inpaint, dict) else None
            top = self._Leaf(**{**tree.__dict__, "unet": self.unet, "enhanced": True})       # EnhancedInpaintMode on `unet`
            tree = (tree, top, H.GraftUnets, {"blend": blend})
        # engine defaults: unified_pipeline.py:1368-1373 (on, threshold 3.33 %, oos 0.6, 1.0 with an init image)
        if hires_fix is None:
            hires_fix = self.hires_fix
        if hires_oos_fraction is None:
            hires_oos_fraction = self.hires_image_oos_fraction if image is not None else self.hires_oos_fraction
        sample_size = getattr(self.unet.config, "sample_size", 64)
        natural_px = sample_size * self.vae_scale_factor
        use_hires = False
        if hires_fix and not (width < natural_px or height < natural_px):
            threshold = math.floor(natural_px * (1 + self.hires_threshold_fraction))
            use_hires = not (width <= threshold and height <= threshold)
        if use_hires and not is_k:
            raise ValueError("Can't use Diffuser schedulers with Hires fix. "
                             "Either use a K-Diffusion scheduler or disable Hires fix.")
        if use_hires:
            to_nat = lambda t: None if t is None else H.image_to_natural(natural_px, t if t.ndim == 4 else t[None],
                                                                         hires_oos_fraction)
            nat_image, nat_mask = to_nat(image), to_nat(mask_image)
            natural = self._map_leaves(tree, lambda l: self._Leaf(**{**l.__dict__, "height": natural_px, "width": natural_px,
                                                                     "image": nat_image, "mask_image": nat_mask}))
            tree = (natural, tree, H.HiresUnetWrapper,
                    {"natural_size": [sample_size, sample_size], "oos_fraction": hires_oos_fraction})
        leaves = self._leaves(tree)
        if not is_k and len(leaves) > 1 and use_hires:
            raise ValueError("Can't use Diffuser schedulers with Hires fix.")

        for leaf in leaves:                                     # build_mode: every constructor first
            self._construct_leaf(leaf, generators, B)
        shared = {}
        for leaf in leaves:
            clip_mode = None
            if clip_config is not None:          # one ClipGuidedMode per leaf (:2396-2406), all on the same generators
                fe = self.feature_extractor
                size = fe.size["shortest_edge"] if isinstance(fe.size, dict) else fe.size
                clip_mode = CG.ClipGuidedMode(scheduler=sched, clip_model=self.clip_model, image_mean=fe.image_mean,
                                              image_std=fe.image_std, clip_size=size,
                                              vae_decode=lambda z: self.vae.decode(z).sample,
                                              vae_scale_factor=self.vae_scale_factor, text_embeddings_clip=clip_text_embeddings,
                                              config=clip_config, generators=generators, latent_scale=self.latent_scale)
            self._bind_leaf(leaf, text_embeddings=text_embeddings, uncond_embeddings=uncond_embeddings,
                            guidance_scale=guidance_scale, cfg_execution=cfg_execution, B=B, added_cond=added_cond,
                            uncond_added_cond=uncond_added_cond, cfg_embeddings=shared, clip_mode=clip_mode)
        sched.set_eps_unets([l.eps_unet for l in leaves])
        sched.set_timesteps(num_inference_steps, strength=strength if image is not None else None,
                            config=S.SchedulerConfig(eta=eta, karras_rho=karras_rho, churn=churn, churn_tmin=churn_tmin,
                                                     churn_tmax=churn_tmax, sigma_min=sigma_min, sigma_max=sigma_max,
                                                     noise_type=scheduler_noise_type or "normal"),
                            prediction_type=prediction_type)

        def _blend(mask, u, orig, nxt):
            it = mask.gt(u).to(nxt.dtype)
            return orig * it + nxt * (1 - it)

        # per-leaf scheduler-side UNets (Mode.wrap_k_unet / wrap_d_unet), then collapse the tree (:2462-2473)
        for i, leaf in enumerate(leaves):
            inner = sched.unets[i]
            if is_k:
                if leaf.blend_orig is None:
                    leaf.s_unet = (lambda inner: lambda x, sigma, u: inner(x, sigma))(inner)
                else:
                    leaf.s_unet = (lambda inner, leaf: lambda x, sigma, u:
                                   _blend(leaf.blend_mask, u, leaf.blend_orig.to(x.dtype), inner(x, sigma)))(inner, leaf)
            else:
                if leaf.blend_orig is None:
                    leaf.s_unet = (lambda inner: lambda x, t, u: inner(x, t))(inner)
                else:
                    leaf.s_unet = (lambda inner, leaf: lambda x, t, u:
                                   _blend(leaf.blend_mask, u, sched.add_noise_at(leaf.blend_orig, leaf.noise, t).to(x.dtype),
                                          inner(x, t)))(inner, leaf)

        for leaf in leaves:                                     # ClipGuidedMode.wrap_k_unet / wrap_d_unet (outermost)
            if leaf.clip_mode is not None:
                leaf.s_unet = leaf.clip_mode.wrap_k_unet(leaf.s_unet) if is_k else leaf.clip_mode.wrap_d_unet(leaf.s_unet)

        def collapse(t):
            if not isinstance(t, tuple):
                return t.s_unet
            kw = {k: v for k, v in t[3].items() if v is not None}
            return t[2](collapse(t[0]), collapse(t[1]), generators, **kw)

        def initial(t):                                       # every leaf generates (and draws), the merger picks
            if not isinstance(t, tuple):
                return self._generate_leaf_latents(t, sched, generators, fill_strength)
            left, right = initial(t[0]), initial(t[1])
            return t[2].merge_initial_latents(left, right)

        def split(t, result):
            if not isinstance(t, tuple):
                return result
            return t[2].split_result(split(t[0], result), split(t[1], result))

        model = collapse(tree)
        latents = initial(tree)
        plain = len(leaves) == 1 and leaves[0].blend_orig is None and (leaves[0].clip_mode is None or not is_k)
        self.last_clip_modes = [l.clip_mode for l in leaves if l.clip_mode is not None]
        if is_k:
            latents = sched.loop(latents, callback=callback, k_model=None if plain else model)
        else:
            latents = sched.loop(latents, callback=callback, d_model=None if plain else model)
        latents = split(tree, latents)
        self.last_unet_evals = sum(u.evals for u in sched.unets)
        if output_type == "latent":
            return latents
        result = self.vae_decode(latents)
        if image is not None and outmask_image is not None:         # unified_pipeline.py:2493-2510
            src = image if image.ndim == 4 else image[None]
            om = outmask_image if outmask_image.ndim == 4 else outmask_image[None]
            result = I.outmask_composite(result, src.to(result.device), om.to(result.device))
        return result
