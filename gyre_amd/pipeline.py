"""Diffusers-free generation pipeline over the native UNet / VAE.

Host orchestration restating the reference's ``UnifiedPipeline.__call__``
(gyre/pipeline/unified_pipeline.py:1722-2538) for the modes on the hot path:

  Txt2imgMode                 unified_pipeline.py:155-237
  Img2imgMode                 unified_pipeline.py:240-337
  EnhancedInpaintMode         unified_pipeline.py:398-645 (per-step blend of original vs predicted latents)
  EnhancedRunwayInpaintMode   unified_pipeline.py:648-696 (9-channel UNet input assembly)
  (strength >= 1 "shaped noise" fill, unified_pipeline.py:466-601, is not implemented)

Call stack per SURVEY.md 3.2/3.3: embeddings -> UNetWithEmbeddings -> (UnetWithExtraChannels)
-> CFGUNet_Parallel -> KDiffusionUNetWrapper -> sampler loop -> vae.decode(latents / 0.18215)
-> (x/2+0.5).clamp(0,1).  Everything except the UNet / VAE forward is thin PyTorch host code.

Random draws are per image (one generator per seed, reference pipeline_wrapper.py:243-253 +
randtools.py:39-64), which is what makes data-parallel sharding exact: an image's result
does not depend on which other images share its batch or GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

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
    latent_scale = 0.18215

    def __init__(self, unet, vae, text_encoder: Optional[Callable] = None, device="cuda:0"):
        self.unet, self.vae, self.text_encoder = unet, vae, text_encoder
        self.device = torch.device(device)

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

    # -- the generation call ------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, *, seeds: Sequence[int], text_embeddings: Optional[Tensor] = None,
                 uncond_embeddings: Optional[Tensor] = None, input_ids: Optional[Tensor] = None,
                 negative_ids: Optional[Tensor] = None, height: int = 512, width: int = 512,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, sampler: str = "dpmpp_2m",
                 image: Optional[Tensor] = None, mask_image: Optional[Tensor] = None, strength: float = 0.8,
                 karras_rho: Optional[float] = None, eta: Optional[float] = None, cfg_execution: str = "parallel",
                 output_type: str = "image", callback=None, generator_device: str = "cpu"):
        if height % self.vae_scale_factor or width % self.vae_scale_factor:
            raise ValueError(f"`height` and `width` have to be divisible by {self.vae_scale_factor} "
                             f"but are {height} and {width}.")
        B = len(seeds)
        if B < 1:
            raise ValueError("at least one seed (image) is required")
        dev = self.device
        generators = build_generators(seeds, generator_device)
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

        in_ch = self.unet.config.in_channels
        runway = in_ch == 9
        if runway and (image is None or mask_image is None):
            raise ValueError("the 9-channel inpaint UNet needs image and mask_image")
        lat_ch = 4
        lat_h, lat_w = height // self.vae_scale_factor, width // self.vae_scale_factor

        sched = S.make_scheduler(sampler, generators, dev, torch.float32)
        extra = None
        init_latents = None
        blend_orig = blend_mask = None
        if image is not None:
            if not 0 <= strength <= 1:
                raise NotImplementedError("strength outside [0,1] (shaped-noise fill) is not on the native path yet")
            img = self.preprocess_image(image.to(torch.float32))
            if img.shape[-2:] != (height, width):
                raise ValueError(f"image is {tuple(img.shape[-2:])}, expected {(height, width)}")
            if mask_image is not None:
                mask = self.preprocess_mask(mask_image.to(torch.float32)).to(dev)      # 1 keep / 0 replace
                high_mask = round_mask(mask, 0.001)
                orig = self.image_to_latents(img, generators, high_mask)
                latent_mask = torch.cat([mask_to_latent_mask(mask)] * B)
                if runway:
                    inpaint_mask = 1 - round_mask(latent_mask, 0.001)[:, [0]]           # 0 keep / 1 replace
                    extra = torch.cat([inpaint_mask, orig], dim=1)
                init_latents = self.image_to_latents(img, generators)
                if not runway:
                    # EnhancedInpaintMode: keep the protected area pinned to the (masked) original by blending the
                    # denoised prediction with it while the blend mask exceeds the progress u (_blend, :620-625)
                    blend_orig, blend_mask = orig, latent_mask
            else:
                init_latents = self.image_to_latents(img, generators)

        # ---- UNet stack: embeddings -> extra channels -> CFG -> k-diffusion denoiser -------------------
        def bind(emb):
            u = S.UNetWithEmbeddings(self.unet, emb)
            return S.UnetWithExtraChannels(u, extra) if extra is not None else u

        if do_cfg:
            if cfg_execution == "sequential":
                eps_unet = S.CFGUNet_Sequential(bind(text_embeddings), bind(uncond_embeddings), guidance_scale, B)
            else:
                eps_unet = S.CFGUNet_Parallel(bind(torch.cat([uncond_embeddings, text_embeddings])), guidance_scale, B)
        else:
            eps_unet = bind(text_embeddings)
        sched.set_eps_unet(eps_unet)
        sched.set_timesteps(num_inference_steps, strength=strength if image is not None else None,
                            config=S.SchedulerConfig(eta=eta, karras_rho=karras_rho))

        if init_latents is None:
            sample_size = getattr(self.unet.config, "sample_size", 64)
            latents = txt2img_latents(generators, lat_ch, lat_h, lat_w, sample_size, dev)
            latents = sched.prepare_initial_latents(latents)
        else:
            noise = S.batched_randn(init_latents.shape, generators, dev, torch.float32)
            latents = sched.add_noise(init_latents, noise)

        wrap = {}
        if blend_orig is not None:
            def _blend(u, orig, nxt):
                it = blend_mask.gt(u).to(nxt.dtype)
                return orig * it + nxt * (1 - it)
            if isinstance(sched, S.KDiffusionScheduler):
                wrap["k_wrap"] = lambda px0, u: _blend(u, blend_orig.to(px0.dtype), px0)
            else:
                wrap["d_wrap"] = lambda xt, t, u: _blend(u, sched.add_noise_at(blend_orig, noise, t).to(xt.dtype), xt)
        latents = sched.loop(latents, callback=callback, **wrap)
        self.last_unet_evals = sched.unet.evals
        if output_type == "latent":
            return latents
        return self.vae_decode(latents)
