"""fp32 CPU restatement of the UNet / VAE the reference calls across its
drop-in boundary.  TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED: the arithmetic lives in diffusers ~= 0.16.0
(reference pyproject.toml:22), absent from /root/reference; call sites are
  gyre/pipeline/unet/core.py:262-274      unet(latents, t, encoder_hidden_states=...).sample
  gyre/pipeline/unified_pipeline.py:309   vae.encode(image).latent_dist
  gyre/pipeline/unified_pipeline.py:1531  vae.decode(latents).sample
Topology / hyper-parameters follow gyre/ldm_config/v1-inference.yaml:29-64 and
the diffusers key names consumed by gyre/ckpt_utils.py:259-285.
Pinned as far as the tree allows: hyper-parameters and encoder wiring against the vendored ControlNet copy
(tests/test_oracle_structure.py), and the UNet ASSEMBLY by executing the reference's vendored block forwards
(nonfree/tome_unet.py:34-221, models/memory_efficient_cross_attention.py:32-60) over this file's leaf functions
(tests/test_oracle_block_wiring.py: every level equal to 1e-5).  The leaves' internals and the VAE stay unpinned.

Everything is plain ``torch.nn.functional`` over a flat ``{diffusers_key: tensor}``
dict, i.e. the same ATen ops the reference's CPU path dispatches to
(conv2d, group_norm, linear, baddbmm+softmax, layer_norm, gelu, silu,
interpolate) - which is why this file doubles as the reported CPU baseline.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# configs (mirrors of diffusers config.json fields the reference reads:
# unified_pipeline.py:186 in_channels, :1318 sample_size, :1402 block_out_channels)
# --------------------------------------------------------------------------
@dataclass
class UNetRefConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # per level: does the level carry Transformer2D blocks
    attn_levels: Tuple[bool, ...] = (True, True, True, False)
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)  # SD1.x "attention_head_dim: 8" is really num_heads
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    transformer_depth: Tuple[int, ...] = (1, 1, 1, 1)
    use_linear_projection: bool = False
    sample_size: int = 64
    flip_sin_to_cos: bool = True
    freq_shift: float = 0.0


@dataclass
class VAERefConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def _gn(x: Tensor, sd: SD, p: str, groups: int, eps: float) -> Tensor:
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


# Circular ("tiling") convolutions, reference gyre/pipeline/unified_pipeline.py:1671-1694 (set_module_tiling): the module's own
# padding is applied with F.pad(mode="circular") along x (tiling != "y") and / or y (tiling != "x"), zeros on the other axis, and
# the convolution itself runs unpadded.  TILING is test infrastructure state: set through `tiling(mode)`.
TILING = None


class tiling:
    """with tiling("xy" | "x" | "y" | True): every _conv below pads circularly as the reference's patched Conv2d does."""

    def __init__(self, mode):
        self.mode = "xy" if mode is True else mode

    def __enter__(self):
        global TILING
        self.prev, TILING = TILING, (self.mode or None)
        return self

    def __exit__(self, *exc):
        global TILING
        TILING = self.prev
        return False


def _conv(x: Tensor, sd: SD, p: str, stride: int = 1, padding: int = 1) -> Tensor:
    if TILING and padding:
        x = F.pad(x, (padding, padding, 0, 0), mode="circular" if TILING != "y" else "constant")
        x = F.pad(x, (0, 0, padding, padding), mode="circular" if TILING != "x" else "constant")
        padding = 0
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _lin(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def timestep_embedding(t: Tensor, dim: int, flip_sin_to_cos: bool = True, freq_shift: float = 0.0) -> Tensor:
    """diffusers.models.embeddings.get_timestep_embedding [3P]; SD uses
    flip_sin_to_cos=True, downscale_freq_shift=0 -> [cos, sin] order."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def resnet_block(x: Tensor, temb: Optional[Tensor], sd: SD, p: str, groups: int, eps: float) -> Tensor:
    """ResnetBlock2D [3P]: GN+SiLU+conv3x3 (+temb) GN+SiLU+conv3x3 + shortcut."""
    h = F.silu(_gn(x, sd, p + ".norm1", groups, eps))
    h = _conv(h, sd, p + ".conv1")
    if temb is not None and (p + ".time_emb_proj.weight") in sd:
        h = h + _lin(F.silu(temb), sd, p + ".time_emb_proj")[:, :, None, None]
    h = F.silu(_gn(h, sd, p + ".norm2", groups, eps))
    h = _conv(h, sd, p + ".conv2")
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(x, sd, p + ".conv_shortcut", padding=0)
    return x + h


def attention(q: Tensor, k: Tensor, v: Tensor, heads: int) -> Tensor:
    """[B,N,h*d] -> [B*h,N,d], softmax(q k^T * d^-1/2) v, no mask.  Same reshape /
    scale as reference gyre/pipeline/models/memory_efficient_cross_attention.py:32-60."""
    B, Nq, C = q.shape
    d = C // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(B * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    scores = torch.baddbmm(
        torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
        q, k.transpose(1, 2), beta=0, alpha=d ** -0.5)
    probs = scores.softmax(dim=-1)
    out = torch.bmm(probs, v)
    return out.reshape(B, heads, Nq, d).permute(0, 2, 1, 3).reshape(B, Nq, C)


def basic_transformer_block(x: Tensor, ctx: Tensor, sd: SD, p: str, heads: int, tome_r: int = 0) -> Tensor:
    """BasicTransformerBlock [3P]: LN+self-attn, LN+cross-attn, LN+GEGLU-FF (erf gelu).
    tome_r > 0: ToMe on the self-attention's keys / values (reference nonfree/tome_unet.py:138-182; oracle/tome_ref.py)."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    k1, v1 = _lin(h, sd, p + ".attn1.to_k"), _lin(h, sd, p + ".attn1.to_v")
    if tome_r > 0 and h.shape[1] % 16 == 0:           # the native path merges when the token count is a multiple of 16
        from . import tome_ref
        k1, v1 = tome_ref.tome_merge_kv(k1, v1, tome_r)
    a = attention(_lin(h, sd, p + ".attn1.to_q"), k1, v1, heads)
    x = x + _lin(a, sd, p + ".attn1.to_out.0")
    h = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    a = attention(_lin(h, sd, p + ".attn2.to_q"), _lin(ctx, sd, p + ".attn2.to_k"),
                  _lin(ctx, sd, p + ".attn2.to_v"), heads)
    x = x + _lin(a, sd, p + ".attn2.to_out.0")
    h = F.layer_norm(x, (C,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5)
    h = _lin(h, sd, p + ".ff.net.0.proj")
    val, gate = h.chunk(2, dim=-1)
    h = val * F.gelu(gate)
    x = x + _lin(h, sd, p + ".ff.net.2")
    return x


def transformer_2d(x: Tensor, ctx: Tensor, sd: SD, p: str, heads: int, groups: int, depth: int,
                   linear_proj: bool, tome_r: int = 0) -> Tensor:
    """Transformer2DModel [3P]: GN(eps 1e-6), proj_in, blocks, proj_out, + residual."""
    B, C, H, W = x.shape
    res = x
    h = _gn(x, sd, p + ".norm", groups, 1e-6)
    if not linear_proj:
        h = _conv(h, sd, p + ".proj_in", padding=0)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    else:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = _lin(h, sd, p + ".proj_in")
    for d in range(depth):
        h = basic_transformer_block(h, ctx, sd, f"{p}.transformer_blocks.{d}", heads, tome_r)
    if not linear_proj:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = _conv(h, sd, p + ".proj_out", padding=0)
    else:
        h = _lin(h, sd, p + ".proj_out")
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + res


# --------------------------------------------------------------------------
# UNet2DConditionModel forward
# --------------------------------------------------------------------------
def added_cond_embedding(sd: SD, cfg, text_embeds: Tensor, time_ids: Tensor) -> Tensor:
    """SDXL `text_time` conditioning [3P diffusers UNet2DConditionModel.get_aug_embed]: sinusoid(time_ids) ++ pooled
    text -> add_embedding MLP.  Not part of the reference (SDXL is an extension)."""
    te = timestep_embedding(time_ids.flatten(), cfg.addition_time_embed_dim, cfg.flip_sin_to_cos, cfg.freq_shift)
    te = te.reshape(text_embeds.shape[0], -1)
    a = torch.cat([text_embeds, te.to(text_embeds.dtype)], dim=-1)
    a = _lin(a, sd, "add_embedding.linear_1")
    return _lin(F.silu(a), sd, "add_embedding.linear_2")


def unet_forward(sd: SD, cfg: UNetRefConfig, latents: Tensor, t: Tensor, encoder_hidden_states: Tensor,
                 taps: Optional[dict] = None, added_cond: Optional[dict] = None, tome_r: int = 0,
                 down_res: Optional[Sequence[Tensor]] = None, mid_res: Optional[Tensor] = None,
                 adapter_states: Optional[Sequence[Tensor]] = None) -> Tensor:
    """eps = unet(latents[NCHW], t[int64 N], ctx[N,S,D]).  ``taps`` (optional dict)
    receives named intermediate activations for block-level parity tests.
    down_res / mid_res: ControlNet residuals with the semantics of the reference's in-tree patcher
    (gyre/pipeline/controlnet/unet_patcher.py:30-95): added to the skip connections as the UP path consumes them
    (`res_sample + extra`) and to the mid block's output; the down path and the mid block see the plain activations.
    adapter_states: T2I-adapter features, one per down level, semantics of gyre/pipeline/t2i_adapter/unet_patcher.py:21-86 -
    `hidden_states += adapter_state` in place on the level's last hidden state: before the downsampler on cross-attention
    levels (the tensor is also the skip connection just stored), after the whole level otherwise."""
    g, eps = cfg.norm_num_groups, 1e-5
    boc = cfg.block_out_channels
    if t.ndim == 0:
        t = t[None].expand(latents.shape[0])
    temb = timestep_embedding(t, boc[0], cfg.flip_sin_to_cos, cfg.freq_shift).to(latents.dtype)
    temb = _lin(temb, sd, "time_embedding.linear_1")
    temb = _lin(F.silu(temb), sd, "time_embedding.linear_2")
    if added_cond is not None:
        temb = temb + added_cond_embedding(sd, cfg, added_cond["text_embeds"], added_cond["time_ids"])
    if taps is not None:
        taps["temb"] = temb

    h = _conv(latents, sd, "conv_in")
    skips: List[Tensor] = [h]
    nlev = len(boc)
    for i in range(nlev):
        for j in range(cfg.layers_per_block):
            h = resnet_block(h, temb, sd, f"down_blocks.{i}.resnets.{j}", g, eps)
            if cfg.attn_levels[i]:
                h = transformer_2d(h, encoder_hidden_states, sd, f"down_blocks.{i}.attentions.{j}",
                                   cfg.num_heads[i], g, cfg.transformer_depth[i], cfg.use_linear_projection, tome_r)
            skips.append(h)
        adapt_before = adapter_states is not None and (cfg.attn_levels[i] or i == nlev - 1)
        if adapt_before:                         # in place in the reference: the stored skip connection changes with it
            h = h + adapter_states[i]
            skips[-1] = h
        if i < nlev - 1:
            h = _conv(h, sd, f"down_blocks.{i}.downsamplers.0.conv", stride=2, padding=1)
            skips.append(h)
        if adapter_states is not None and not adapt_before:
            h = h + adapter_states[i]
            skips[-1] = h
        if taps is not None:
            taps[f"down{i}"] = h

    h = resnet_block(h, temb, sd, "mid_block.resnets.0", g, eps)
    h = transformer_2d(h, encoder_hidden_states, sd, "mid_block.attentions.0", cfg.num_heads[-1], g,
                       cfg.transformer_depth[-1], cfg.use_linear_projection, tome_r)
    h = resnet_block(h, temb, sd, "mid_block.resnets.1", g, eps)
    if taps is not None:
        taps["mid"] = h
    if down_res is not None:
        if len(down_res) != len(skips):
            raise ValueError(f"{len(skips)} down-block residuals expected, got {len(down_res)}")
        skips = [s_ + r for s_, r in zip(skips, down_res)]
    if mid_res is not None:
        h = h + mid_res

    for i in range(nlev):
        lvl = nlev - 1 - i
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(h, temb, sd, f"up_blocks.{i}.resnets.{j}", g, eps)
            if cfg.attn_levels[lvl]:
                h = transformer_2d(h, encoder_hidden_states, sd, f"up_blocks.{i}.attentions.{j}",
                                   cfg.num_heads[lvl], g, cfg.transformer_depth[lvl], cfg.use_linear_projection, tome_r)
        if i < nlev - 1:
            # diffusers resizes to the next skip connection's size (forward_upsample_size) - 2x unless a level is odd
            h = F.interpolate(h, size=skips[-1].shape[-2:], mode="nearest")
            h = _conv(h, sd, f"up_blocks.{i}.upsamplers.0.conv")
        if taps is not None:
            taps[f"up{i}"] = h

    h = F.silu(_gn(h, sd, "conv_norm_out", g, eps))
    return _conv(h, sd, "conv_out")


# --------------------------------------------------------------------------
# AutoencoderKL
# --------------------------------------------------------------------------
def _vae_attn(x: Tensor, sd: SD, p: str, groups: int) -> Tensor:
    """diffusers 0.16 AttentionBlock (1 head, d=C) [3P].  Accepts both the 0.16 key
    names (group_norm/query/key/value/proj_attn) and the later to_q/... names."""
    B, C, H, W = x.shape
    names = ("group_norm", "query", "key", "value", "proj_attn")
    if (p + ".to_q.weight") in sd:
        names = ("group_norm", "to_q", "to_k", "to_v", "to_out.0")
    h = _gn(x, sd, f"{p}.{names[0]}", groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    q, k, v = (_lin(h, sd, f"{p}.{n}") for n in names[1:4])
    a = attention(q, k, v, 1)
    a = _lin(a, sd, f"{p}.{names[4]}")
    return x + a.reshape(B, H, W, C).permute(0, 3, 1, 2)


def _vae_mid(h: Tensor, sd: SD, p: str, groups: int) -> Tensor:
    h = resnet_block(h, None, sd, p + ".resnets.0", groups, 1e-6)
    h = _vae_attn(h, sd, p + ".attentions.0", groups)
    return resnet_block(h, None, sd, p + ".resnets.1", groups, 1e-6)


def vae_encode_moments(sd: SD, cfg: VAERefConfig, image: Tensor) -> Tensor:
    """image [B,3,H,W] in -1..1 -> moments [B,2*z,H/8,W/8] (mean | logvar)."""
    g = cfg.norm_num_groups
    boc = cfg.block_out_channels
    h = _conv(image, sd, "encoder.conv_in")
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block):
            h = resnet_block(h, None, sd, f"encoder.down_blocks.{i}.resnets.{j}", g, 1e-6)
        if i < len(boc) - 1:
            h = F.pad(h, (0, 1, 0, 1))  # asymmetric pad, stride-2 conv pad 0 [3P Downsample2D]
            h = _conv(h, sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
    h = _vae_mid(h, sd, "encoder.mid_block", g)
    h = F.silu(_gn(h, sd, "encoder.conv_norm_out", g, 1e-6))
    h = _conv(h, sd, "encoder.conv_out")
    return _conv(h, sd, "quant_conv", padding=0)


def vae_posterior_sample(moments: Tensor, generator: Optional[torch.Generator]) -> Tensor:
    """DiagonalGaussianDistribution.sample [3P]: mean + exp(0.5*clamp(logvar,-30,20)) * randn."""
    mean, logvar = moments.chunk(2, dim=1)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    noise = torch.randn(mean.shape, generator=generator, dtype=mean.dtype,
                        device=generator.device if generator is not None else mean.device)
    return mean + std * noise.to(mean.device)


def vae_decode(sd: SD, cfg: VAERefConfig, z: Tensor) -> Tensor:
    """latents [B,4,h,w] (already divided by the scaling factor) -> image [B,3,8h,8w]."""
    g = cfg.norm_num_groups
    boc = list(reversed(cfg.block_out_channels))
    h = _conv(z, sd, "post_quant_conv", padding=0)
    h = _conv(h, sd, "decoder.conv_in")
    h = _vae_mid(h, sd, "decoder.mid_block", g)
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block + 1):
            h = resnet_block(h, None, sd, f"decoder.up_blocks.{i}.resnets.{j}", g, 1e-6)
        if i < len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, sd, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    h = F.silu(_gn(h, sd, "decoder.conv_norm_out", g, 1e-6))
    return _conv(h, sd, "decoder.conv_out")
