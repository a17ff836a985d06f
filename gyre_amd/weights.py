"""Parameter inventory (diffusers key names + shapes) and seeded synthetic weights.

State-dict keys equal the checkpoint's diffusers names because the reference's
loader falls back to ``Class(**config)`` + ``load_state_dict`` of the single
``*.safetensors`` in the model dir (reference gyre/manager.py:1068-1112), and
``ckpt_utils.py:259-285`` converts ``.ckpt`` files into that key space.

There are no SD weights on the build/bench machines (no network), so benches
and parity tests use ``synthetic_state_dict``: every tensor is drawn from its
own generator seeded by ``seed ^ crc32(key)`` so the values do not depend on
enumeration order.  Fan-in scaling keeps activations O(1) through the whole
network so that parity errors are meaningful (a N(0, 0.02) init would collapse
the signal to zero after a few layers and hide kernel bugs).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import UNetConfig, VAEConfig

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def _resnet(shapes, p, cin, cout, temb_dim):
    shapes[p + ".norm1.weight"] = (cin,)
    shapes[p + ".norm1.bias"] = (cin,)
    shapes[p + ".conv1.weight"] = (cout, cin, 3, 3)
    shapes[p + ".conv1.bias"] = (cout,)
    if temb_dim:
        shapes[p + ".time_emb_proj.weight"] = (cout, temb_dim)
        shapes[p + ".time_emb_proj.bias"] = (cout,)
    shapes[p + ".norm2.weight"] = (cout,)
    shapes[p + ".norm2.bias"] = (cout,)
    shapes[p + ".conv2.weight"] = (cout, cout, 3, 3)
    shapes[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        shapes[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        shapes[p + ".conv_shortcut.bias"] = (cout,)


def _transformer(shapes, p, c, ctx_dim, depth, linear_proj):
    shapes[p + ".norm.weight"] = (c,)
    shapes[p + ".norm.bias"] = (c,)
    shapes[p + ".proj_in.weight"] = (c, c) if linear_proj else (c, c, 1, 1)
    shapes[p + ".proj_in.bias"] = (c,)
    for d in range(depth):
        b = f"{p}.transformer_blocks.{d}"
        for n in ("norm1", "norm2", "norm3"):
            shapes[f"{b}.{n}.weight"] = (c,)
            shapes[f"{b}.{n}.bias"] = (c,)
        for a, kv in (("attn1", c), ("attn2", ctx_dim)):
            shapes[f"{b}.{a}.to_q.weight"] = (c, c)
            shapes[f"{b}.{a}.to_k.weight"] = (c, kv)
            shapes[f"{b}.{a}.to_v.weight"] = (c, kv)
            shapes[f"{b}.{a}.to_out.0.weight"] = (c, c)
            shapes[f"{b}.{a}.to_out.0.bias"] = (c,)
        shapes[f"{b}.ff.net.0.proj.weight"] = (8 * c, c)
        shapes[f"{b}.ff.net.0.proj.bias"] = (8 * c,)
        shapes[f"{b}.ff.net.2.weight"] = (c, 4 * c)
        shapes[f"{b}.ff.net.2.bias"] = (c,)
    shapes[p + ".proj_out.weight"] = (c, c) if linear_proj else (c, c, 1, 1)
    shapes[p + ".proj_out.bias"] = (c,)


def unet_param_shapes(cfg: UNetConfig) -> Shapes:
    s: Shapes = OrderedDict()
    boc = cfg.block_out_channels
    td = cfg.time_embed_dim
    s["time_embedding.linear_1.weight"] = (td, boc[0])
    s["time_embedding.linear_1.bias"] = (td,)
    s["time_embedding.linear_2.weight"] = (td, td)
    s["time_embedding.linear_2.bias"] = (td,)
    if getattr(cfg, "addition_embed_type", None) == "text_time":
        s["add_embedding.linear_1.weight"] = (td, cfg.projection_class_embeddings_input_dim)
        s["add_embedding.linear_1.bias"] = (td,)
        s["add_embedding.linear_2.weight"] = (td, td)
        s["add_embedding.linear_2.bias"] = (td,)
    s["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (boc[0],)
    n = len(boc)
    skip_ch = [boc[0]]
    cin = boc[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            _resnet(s, f"down_blocks.{i}.resnets.{j}", cin, boc[i], td)
            cin = boc[i]
            if cfg.attn_levels[i]:
                _transformer(s, f"down_blocks.{i}.attentions.{j}", cin, cfg.cross_attention_dim,
                             cfg.transformer_depth[i], cfg.use_linear_projection)
            skip_ch.append(cin)
        if i < n - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (cin, cin, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (cin,)
            skip_ch.append(cin)
    _resnet(s, "mid_block.resnets.0", cin, cin, td)
    _transformer(s, "mid_block.attentions.0", cin, cfg.cross_attention_dim, cfg.transformer_depth[-1],
                 cfg.use_linear_projection)
    _resnet(s, "mid_block.resnets.1", cin, cin, td)
    for i in range(n):
        lvl = n - 1 - i
        for j in range(cfg.layers_per_block + 1):
            sk = skip_ch.pop()
            _resnet(s, f"up_blocks.{i}.resnets.{j}", cin + sk, boc[lvl], td)
            cin = boc[lvl]
            if cfg.attn_levels[lvl]:
                _transformer(s, f"up_blocks.{i}.attentions.{j}", cin, cfg.cross_attention_dim,
                             cfg.transformer_depth[lvl], cfg.use_linear_projection)
        if i < n - 1:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (cin, cin, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (cin,)
    s["conv_norm_out.weight"] = (cin,)
    s["conv_norm_out.bias"] = (cin,)
    s["conv_out.weight"] = (cfg.out_channels, cin, 3, 3)
    s["conv_out.bias"] = (cfg.out_channels,)
    return s


def _vae_attn(s, p, c):
    # diffusers 0.16 AttentionBlock names (reference pins diffusers ~= 0.16.0)
    s[p + ".group_norm.weight"] = (c,)
    s[p + ".group_norm.bias"] = (c,)
    for n in ("query", "key", "value", "proj_attn"):
        s[f"{p}.{n}.weight"] = (c, c)
        s[f"{p}.{n}.bias"] = (c,)


def vae_param_shapes(cfg: VAEConfig, encoder: bool = True, decoder: bool = True) -> Shapes:
    s: Shapes = OrderedDict()
    boc = cfg.block_out_channels
    z = cfg.latent_channels
    if encoder:
        s["encoder.conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
        s["encoder.conv_in.bias"] = (boc[0],)
        cin = boc[0]
        for i, c in enumerate(boc):
            for j in range(cfg.layers_per_block):
                _resnet(s, f"encoder.down_blocks.{i}.resnets.{j}", cin, c, 0)
                cin = c
            if i < len(boc) - 1:
                s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
                s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
        _resnet(s, "encoder.mid_block.resnets.0", cin, cin, 0)
        _vae_attn(s, "encoder.mid_block.attentions.0", cin)
        _resnet(s, "encoder.mid_block.resnets.1", cin, cin, 0)
        s["encoder.conv_norm_out.weight"] = (cin,)
        s["encoder.conv_norm_out.bias"] = (cin,)
        s["encoder.conv_out.weight"] = (2 * z, cin, 3, 3)
        s["encoder.conv_out.bias"] = (2 * z,)
        s["quant_conv.weight"] = (2 * z, 2 * z, 1, 1)
        s["quant_conv.bias"] = (2 * z,)
    if decoder:
        rb = list(reversed(boc))
        s["post_quant_conv.weight"] = (z, z, 1, 1)
        s["post_quant_conv.bias"] = (z,)
        s["decoder.conv_in.weight"] = (rb[0], z, 3, 3)
        s["decoder.conv_in.bias"] = (rb[0],)
        cin = rb[0]
        _resnet(s, "decoder.mid_block.resnets.0", cin, cin, 0)
        _vae_attn(s, "decoder.mid_block.attentions.0", cin)
        _resnet(s, "decoder.mid_block.resnets.1", cin, cin, 0)
        for i, c in enumerate(rb):
            for j in range(cfg.layers_per_block + 1):
                _resnet(s, f"decoder.up_blocks.{i}.resnets.{j}", cin, c, 0)
                cin = c
            if i < len(rb) - 1:
                s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
                s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
        s["decoder.conv_norm_out.weight"] = (cin,)
        s["decoder.conv_norm_out.bias"] = (cin,)
        s["decoder.conv_out.weight"] = (cfg.out_channels, cin, 3, 3)
        s["decoder.conv_out.bias"] = (cfg.out_channels,)
    return s


# branch-closing layers get a smaller gain so the residual stream stays O(1)
_BRANCH_OUT = (".conv2.weight", ".to_out.0.weight", ".ff.net.2.weight", ".proj_out.weight", ".proj_attn.weight")


def synthetic_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed((seed * 1000003) ^ zlib.crc32(key.encode()))
    is_norm = ".norm" in key or "group_norm" in key or key.startswith("conv_norm_out") or "conv_norm_out" in key
    if key.endswith(".bias"):
        std = 0.1 if is_norm else 0.05
        return torch.randn(shape, generator=g) * std
    if is_norm:
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    gain = 0.5 if key.endswith(_BRANCH_OUT) else 1.0
    return torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)


def synthetic_state_dict(shapes: Shapes, seed: int = 0) -> Dict[str, torch.Tensor]:
    return OrderedDict((k, synthetic_tensor(k, shp, seed)) for k, shp in shapes.items())
