"""CLIP guidance of the denoising loop (host side).

Restates the reference's ``ClipGuidedMode`` (gyre/pipeline/unet/clipguided.py:94-420), ``MakeCutouts`` (:38-85),
``spherical_dist_loss`` (:88-91), ``ClipGuidanceConfig`` (:26-35) and ``VaeApproximator``
(gyre/pipeline/vae_approximator.py:4-43):

  every step (until the loss trend is flat, :153-173) the guided UNet evaluation is repeated under autograd from
  ``latents.detach().requires_grad_()``; the predicted clean sample is turned into images - VAE-decoded cut-outs and / or
  cut-outs of a 4x3 linear "approximate decode" - embedded by CLIP, and  -d(500 * scale * loss)/d latents  is added to the
  denoised prediction (k-diffusion samplers: ``+ grads * sigma**2``, :297) or to the noise prediction (diffusers-style
  samplers: ``- sqrt(1 - alpha_bar_t) * grads``, :212-216).

The autograd graph here is thin PyTorch glue (scaling, cropping, the resampling matrices of gyre_amd/resize.py, the loss);
the two heavy nodes are the native UNet and VAE decoder, whose backward is libgyre_hip's input-gradient sweep
(gyre_unet_vjp / gyre_vae_decode_vjp through the autograd nodes in gyre_amd/modules.py).  The CLIP model itself is a
PyTorch module supplied by the caller (north_star: text / image encoders stay host PyTorch).

Reference quirks that are kept, because they change results:
  * with BOTH approx_cutouts and vae_cutouts > 0 (the default 2 + 2) the decoded VAE cut-outs are overwritten by a view of
    the approximate ones (``image2 = image.view(...)``, :396) - the random draws for them still happen, the decode does
    not influence the loss; the two counts must then be equal or the view fails;
  * the loss history divides by the batch size, the flat-loss test is batch-wide (:418, :153-173), so guidance is the one
    mode whose images depend on their batch.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import schedulers as S
from .resize import resize_right

Tensor = torch.Tensor


@dataclass
class ClipGuidanceConfig:                       # clipguided.py:26-35
    guidance_scale: float = 0
    guidance_base: str = "guided"               # "guided" | "mixed"
    gradient_length: int = 15
    gradient_threshold: float = 0.01
    gradient_maxloss: float = 1.0
    vae_cutouts: int = 2
    approx_cutouts: int = 2
    no_cutouts: object = False                  # False | True | "vae" | "approx"


class VaeApproximator:
    """latents -> approximate RGB by a fixed 4x3 matrix (vae_approximator.py:4-43; values for SD 1.4)."""

    FACTORS = ((0.298, 0.207, 0.208), (0.187, 0.286, 0.173), (-0.158, 0.189, 0.264), (-0.184, -0.271, -0.473))

    def __init__(self, device=None, dtype=None):
        self.latent_rgb_factors = torch.tensor(self.FACTORS, dtype=dtype, device=device)

    def __call__(self, latents: Tensor) -> Tensor:
        self.latent_rgb_factors = self.latent_rgb_factors.to(latents.device, latents.dtype)
        return torch.einsum("...lhw,lr -> ...rhw", latents, self.latent_rgb_factors)


class MakeCutouts:
    """Random square crops resized to cut_size, per image with that image's generator (clipguided.py:38-85): output is
    grouped by image (b1c1, b1c2, ..., b2c1, ...)."""

    def __init__(self, cut_size: int, generators: Sequence[torch.Generator], cut_power: float = 1.0):
        self.cut_size, self.cut_power, self.generators = cut_size, cut_power, list(generators)

    @staticmethod
    def _randint(high, generator):
        return torch.randint(0, high, (), generator=generator, device=generator.device)

    def __call__(self, pixel_values: Tensor, num_cutouts: int) -> Tensor:
        side_y, side_x = pixel_values.shape[2:4]
        max_size = min(side_x, side_y)
        min_size = min(side_x, side_y, self.cut_size)
        if pixel_values.shape[0] != len(self.generators):
            raise ValueError("one generator per image is required")
        cutouts = []
        for generator, pixels in zip(self.generators, pixel_values.split(1)):
            for _ in range(num_cutouts):
                size = torch.rand([], generator=generator, device=generator.device)
                size = int(size ** self.cut_power * (max_size - min_size) + min_size)
                offsetx = int(self._randint(side_x - size + 1, generator))
                offsety = int(self._randint(side_y - size + 1, generator))
                cutout = pixels[:, :, offsety:offsety + size, offsetx:offsetx + size]
                cutouts.append(resize_right(cutout, out_shape=(self.cut_size, self.cut_size), pad_mode="reflect"))
        return torch.cat(cutouts)


def spherical_dist_loss(x: Tensor, y: Tensor) -> Tensor:
    x = torch.nn.functional.normalize(x, dim=-1)
    y = torch.nn.functional.normalize(y, dim=-1)
    return (x - y).norm(dim=-1).div(2).arcsin().pow(2).mul(2)


def _features(out) -> Tensor:
    """CLIPModel.get_*_features returns the projected embedding (a tensor; newer transformers wrap it in an output)."""
    return out if isinstance(out, torch.Tensor) else out.pooler_output


class PatchEmbedAsMatmul(torch.nn.Module):
    """CLIP's patch embedding (a Conv2d with stride = kernel = patch size) as unfold + matmul: the same arithmetic without a
    MIOpen convolution.  On ROCm the first backward-data call of that one convolution triggers MIOpen's kernel search
    (minutes on a machine with a cold cache); the guidance gradient passes through it every step."""

    def __init__(self, conv: torch.nn.Conv2d):
        super().__init__()
        self.weight, self.bias, self.p = conv.weight, conv.bias, conv.kernel_size[0]

    def forward(self, x: Tensor) -> Tensor:
        B, C, H, W = x.shape
        p, gh, gw = self.p, H // self.p, W // self.p
        # non-overlapping patches are a pure reshape (no im2col kernel): [B, gh*gw, C*p*p] in the conv weight's (c, ky, kx) order
        cols = x[:, :, :gh * p, :gw * p].reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
        out = cols @ self.weight.flatten(1).to(x.dtype).t()
        if self.bias is not None:
            out = out + self.bias.to(x.dtype)
        return out.transpose(1, 2).reshape(B, -1, gh, gw)


def patch_embedding_as_matmul(clip_model):
    """In-place: swap a transformers CLIPModel's vision patch-embedding conv for PatchEmbedAsMatmul; returns the model."""
    emb = clip_model.vision_model.embeddings
    if isinstance(emb.patch_embedding, torch.nn.Conv2d):
        emb.patch_embedding = PatchEmbedAsMatmul(emb.patch_embedding)
    return clip_model


def _resize_short_edge(x: Tensor, size: int) -> Tensor:
    """torchvision ``T.Resize(int)`` on a tensor: shorter edge to ``size``, bilinear (the no_cutouts branch, :351-358)."""
    h, w = x.shape[-2:]
    if h <= w:
        nh, nw = size, int(size * w / h)
    else:
        nh, nw = int(size * h / w), size
    return torch.nn.functional.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False)


class ClipGuidedMode:
    """Wraps the UNet stack of one mode-tree leaf.  ``scheduler`` is the loop driver (gyre_amd.schedulers), ``vae_decode``
    maps UNSCALED latents (already divided by 0.18215) to an image in [-1, 1] and must be differentiable."""

    def __init__(self, *, scheduler, clip_model, image_mean: Sequence[float], image_std: Sequence[float], clip_size: int,
                 vae_decode: Callable[[Tensor], Tensor], vae_scale_factor: int, text_embeddings_clip: Tensor,
                 config: ClipGuidanceConfig, generators: List[torch.Generator], latent_scale: float = 0.18215):
        self.scheduler, self.clip_model, self.vae_decode = scheduler, clip_model, vae_decode
        self.text_embeddings_clip, self.config, self.generators = text_embeddings_clip, config, list(generators)
        self.vae_scale_factor, self.latent_scale, self.clip_size = vae_scale_factor, latent_scale, int(clip_size)
        self.mean = torch.tensor(list(image_mean), dtype=torch.float32).view(1, -1, 1, 1)
        self.std = torch.tensor(list(image_std), dtype=torch.float32).view(1, -1, 1, 1)
        self.make_cutouts = MakeCutouts(self.clip_size // vae_scale_factor, generators)
        self.make_cutouts_rgb = MakeCutouts(self.clip_size, generators)
        self.approx_decoder = VaeApproximator()
        self._lossavg: List[float] = []
        self._loss_pending: list = []                # (loss tensor on the device, batch) not yet read back, see `lossavg`
        self.flatloss = False
        self._guided_stem_only = False
        self.grad_evals = 0

    @property
    def lossavg(self) -> List[float]:
        """Loss history (clipguided.py:418).  The reference reads the loss back inside cond_fn, i.e. BEFORE the backward pass is
        queued: a host-device synchronisation that leaves the device idle while the host walks the autograd graph of the CLIP
        model (~400 small launches; ~4 ms of every guided step).  Here the loss stays a device scalar until somebody looks at
        the history - the flat-loss test at the start of the NEXT step, by which time this step's backward pass, reverse sweep
        and unconditional evaluation are all queued.  Same values, same order."""
        if self._loss_pending:
            pend, self._loss_pending = self._loss_pending, []
            self._lossavg.extend(float(t) / b for t, b in pend)
        return self._lossavg

    # ---- loss trend (clipguided.py:153-173) ----
    def _has_flatloss(self) -> bool:
        c = self.config
        if not self.flatloss and len(self._lossavg) + len(self._loss_pending) > c.gradient_length:   # (no read-back before it can matter)
            x = np.linspace(0, 1, c.gradient_length)
            X = np.vstack([x, np.ones(len(x))]).T
            y = np.asarray(self.lossavg[-c.gradient_length:])
            try:
                m, b = np.linalg.lstsq(X, y, rcond=None)[0]
                if abs(m) < c.gradient_threshold and b < c.gradient_maxloss:
                    self.flatloss = True
            except np.linalg.LinAlgError:
                pass
        return self.flatloss

    # ---- noise-prediction level (wrap_guidance_unet, :180-265) ----
    def wrap_guidance_unet(self, unet_g, unet_u, child, guidance_scale: float, unet_both=None):
        """unet_g / unet_u: conditional / unconditional noise predictors (x, t) -> eps; child: their plain CFG combination
        (what the wrapped mode would have built).  unet_u None = no classifier-free guidance.
        unet_both: optional predictor over cat[uncond, cond] embeddings (the parallel CFG binding).  The reference evaluates the
        conditional stem under autograd and, once the gradient is there, the unconditional one on the SAME latents and timestep
        (:218-241): with unet_both the two are one call on cat[x, x] - the unconditional half is used detached and only the
        conditional half is differentiated (modules.grad_samples), same values as the two calls."""
        diffusers_style = isinstance(self.scheduler, S.DiffusersScheduler)
        g_cache: list = []

        def fork(latents, t):                            # k-diffusion samplers (:218-241)
            if self._guided_stem_only:
                if unet_both is not None:
                    from .modules import grad_samples
                    n = latents.shape[0]
                    t2 = torch.cat([t, t]) if isinstance(t, torch.Tensor) and t.shape else t
                    with grad_samples(n, n):
                        both = unet_both(torch.cat([latents, latents]), t2)
                    noise_pred_u, noise_pred_g = both.chunk(2)
                    g_cache.append((noise_pred_g, noise_pred_u.detach()))
                    return noise_pred_g
                noise_pred_g = unet_g(latents, t)
                g_cache.append((noise_pred_g, None))
                return noise_pred_g
            if g_cache:
                noise_pred_g, noise_pred_u = g_cache.pop()
                noise_pred_g = noise_pred_g.detach()
                if noise_pred_u is None:
                    noise_pred_u = unet_u(latents, t)
                return noise_pred_u + guidance_scale * (noise_pred_g - noise_pred_u)
            return child(latents, t)

        def ford(latents, t):                            # diffusers-style samplers (:180-216)
            if self.config.guidance_base == "guided":
                noise_pred_u = unet_u(latents, t)
                noise_pred_g, grads = self.model_d_fn(unet_g, latents, t)
                noise_pred = noise_pred_u + guidance_scale * (noise_pred_g - noise_pred_u)
            else:
                noise_pred, grads = self.model_d_fn(child, latents, t)
            alpha_prod_t = float(self.scheduler.sched.alphas_cumprod[int(t)])
            return noise_pred - (1 - alpha_prod_t) ** 0.5 * grads

        if unet_u is None:                               # no CFG: the reference wraps nothing at this level (:2411-2431)
            if diffusers_style:
                raise ValueError("CLIP guidance with a diffusers-style sampler needs classifier-free guidance (guidance_scale > 1)")
            return child
        inner = ford if diffusers_style else fork

        def wrapped(latents, t):
            if self._has_flatloss():
                return child(latents, t)
            return inner(latents, t)

        return wrapped

    # ---- denoiser level (wrap_k_unet, :267-299) ----
    def wrap_k_unet(self, child):
        """child(x, sigma, u) -> denoised prediction (the wrapped mode's k-unet)."""
        def wrapped(latents, sigma, u):
            if self._has_flatloss():
                return child(latents, sigma, u)
            if self.config.guidance_base == "guided":
                self._guided_stem_only = True
                try:
                    _, grads = self.model_k_fn(child, latents, sigma, u)
                finally:
                    self._guided_stem_only = False
                res = child(latents, sigma, u)
            else:
                self._guided_stem_only = False
                res, grads = self.model_k_fn(child, latents, sigma, u)
            # k_utils.append_dims(sigma, ndim): a per-sample sigma vector broadcasts over its own image (:292-296)
            if not isinstance(sigma, torch.Tensor):
                s = sigma
            elif sigma.numel() == res.shape[0] and sigma.numel() > 1:
                s = sigma.reshape(-1, *[1] * (res.ndim - 1)).to(res.device, res.dtype)
            elif sigma.device.type == "cpu":
                s = float(sigma.reshape(-1)[0])        # host scalar: no host-to-device copy (a synchronising one from pageable memory)
            else:
                s = sigma.reshape(-1)[0].to(res.device, res.dtype)
            return res + grads * (s ** 2)
        return wrapped

    def wrap_d_unet(self, child):
        return child

    # ---- the differentiated evaluations (:301-338) ----
    def model_k_fn(self, unet, latents, sigma, u):
        with torch.enable_grad():
            latents = latents.detach().requires_grad_()
            sample = unet(latents, sigma, u)
            grads = self.cond_fn(latents, sample)
        return sample.detach(), grads

    def model_d_fn(self, unet, latents, t):
        with torch.enable_grad():
            latents = latents.detach().requires_grad_()
            noise_pred = unet(latents, t)
            a = float(self.scheduler.sched.alphas_cumprod[int(t)])
            sample = (latents - (1 - a) ** 0.5 * noise_pred) / a ** 0.5          # DiffusersScheduler.predict_x0
            grads = self.cond_fn(latents, sample)
        return noise_pred.detach(), grads

    # ---- CLIP loss gradient (:340-420) ----
    def cond_fn(self, latents: Tensor, sample: Tensor) -> Tensor:
        c = self.config
        vae_cutouts, approx_cutouts, no_cutouts = c.vae_cutouts, c.approx_cutouts, c.no_cutouts
        num_cutouts = vae_cutouts + approx_cutouts
        batch_total = latents.shape[0]
        with torch.enable_grad():
            if not num_cutouts:
                if no_cutouts == "approx":
                    image = _resize_short_edge(self.approx_decoder(sample), self.clip_size)
                else:
                    sample = _resize_short_edge(sample, self.clip_size // self.vae_scale_factor)
                    image = self.vae_decode(1 / self.latent_scale * sample)
            else:
                image = None
                if approx_cutouts:
                    out_shape = (sample.shape[2] * self.vae_scale_factor, sample.shape[3] * self.vae_scale_factor)
                    image = resize_right(self.approx_decoder(sample), out_shape=out_shape, pad_mode="reflect")
                    image = self.make_cutouts_rgb(image, approx_cutouts)
                if vae_cutouts:
                    sample2 = self.make_cutouts(sample, vae_cutouts)              # draws happen in either branch below
                    if image is None:
                        image = self.vae_decode(1 / self.latent_scale * sample2)
                    else:
                        # reference :392-404: `image2 = image.view(batch_total, vae_cutouts, ...)` - the decoded VAE
                        # cut-outs are replaced by the approximate ones (see module docstring); the decode is skipped here
                        # because its result cannot reach the loss
                        if approx_cutouts != vae_cutouts:
                            raise ValueError("with approx_cutouts and vae_cutouts both > 0 they must be equal "
                                             "(reference clipguided.py:396 views one as the other)")
                        a5 = image.view(batch_total, approx_cutouts, *image.shape[-3:])
                        image = torch.stack([a5, a5], dim=1).view(batch_total * num_cutouts, *image.shape[-3:])
            image = (image / 2 + 0.5).clamp(0, 1)
            if self.mean.device != image.device or self.mean.dtype != image.dtype:     # once: no per-step host-to-device copy
                self.mean, self.std = self.mean.to(image.device, image.dtype), self.std.to(image.device, image.dtype)
            image = (image - self.mean) / self.std
            # the reference hands the CLIP model tensors of the pipeline's own dtype (fp16 end to end on a GPU, :400-404); here the
            # decoded image is fp32, so it is cast to whatever dtype the caller loaded the CLIP model in (no-op for an fp32 model)
            clip_dtype = getattr(self.clip_model, "dtype", None)
            if isinstance(clip_dtype, torch.dtype) and clip_dtype != image.dtype:
                image = image.to(clip_dtype)
            image_embeddings_clip = _features(self.clip_model.get_image_features(image))
            text_clip = self.text_embeddings_clip
            if image_embeddings_clip.dtype != torch.float32:     # reduced-precision CLIP model: the loss itself stays fp32 (2 x B x 512
                image_embeddings_clip, text_clip = image_embeddings_clip.float(), text_clip.float()     # numbers; arcsin / pow in bf16 is noise)
            if no_cutouts:
                loss = spherical_dist_loss(image_embeddings_clip, text_clip).mean()
            else:
                text_in = text_clip.repeat_interleave(num_cutouts, dim=0)
                dists = spherical_dist_loss(image_embeddings_clip, text_in)
                dists = dists.view([num_cutouts, latents.shape[0], -1])
                loss = dists.sum(2).mean(0).sum()
            self._loss_pending.append((loss.detach(), latents.shape[0]))
            self.grad_evals += 1
            return -torch.autograd.grad(loss * (c.guidance_scale * 500), latents)[0]
