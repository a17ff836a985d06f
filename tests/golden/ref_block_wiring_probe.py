"""Build-container probe: the reference's VENDORED copies of diffusers' UNet block forwards (nonfree/tome_unet.py:34-221 -
ToMeDownBlock / ToMeMidBlock / ToMeUpBlock / ToMeSpatialTransformer / ToMeTransformerBlock / ToMeCrossAttention, which re-state
the forward bodies of CrossAttnDownBlock2D, UNetMidBlock2DCrossAttn, CrossAttnUpBlock2D, SpatialTransformer,
BasicTransformerBlock and CrossAttention with r = 0 merging) are EXECUTED from /root/reference with the oracle's LEAF functions
(ResnetBlock, GroupNorm, linear / 1x1 conv, LayerNorm, GEGLU feed-forward, down / up-sampler) plugged in as their sub-modules,
and the assembled trunk is compared with oracle/models_ref.py's own forward.

What this pins: the assembly - order of resnet / attention inside each block, skip-connection tuple bookkeeping and concat order,
mid-block order, transformer residual / reshape / projection order, the three sub-layers of the transformer block, the attention
arithmetic (scale, softmax axis, head split - the head split from gyre/pipeline/models/memory_efficient_cross_attention.py:32-60).
What stays unpinned: the leaves' internals (ResnetBlock2D, Down/Upsample2D, GEGLU FeedForward, timestep embedding) - diffusers
is not vendored.  Prints one JSON object; run by tests/test_oracle_block_wiring.py; nothing here ships."""
import json
import os
import sys
from types import SimpleNamespace

import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    import make_golden as mg
    mg._install()
    sys.path.insert(0, mg.REF)
    from nonfree import tome_unet as T
    from gyre_amd import config as gcfg, weights
    from oracle import models_ref as M
    T.attn_size = None                     # ToMeTransformerBlock.forward reads an undefined global (dead code in the reference)
    cfg = gcfg.tiny_unet()
    sd = weights.synthetic_state_dict(weights.unet_param_shapes(cfg), 0)
    g, eps, boc = cfg.norm_num_groups, 1e-5, cfg.block_out_channels
    out = {}

    def make(cls, **attrs):
        o = object.__new__(cls)
        for k, v in attrs.items():
            object.__setattr__(o, k, v)
        return o

    # ---- CrossAttention (tome_unet.py:138-221) -------------------------------------------------------------------------
    def cross_attention(p, heads):
        d = sd[p + ".to_q.weight"].shape[0] // heads

        def to_batch(t):                     # head split as vendored in memory_efficient_cross_attention.py:32-45
            b = t.shape[0]
            return t.unsqueeze(3).reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)

        def to_heads(t):                     # ... and :52-58
            b = t.shape[0] // heads
            return t.unsqueeze(0).reshape(b, heads, t.shape[1], d).permute(0, 2, 1, 3).reshape(b, t.shape[1], heads * d)
        return make(T.ToMeCrossAttention, to_q=lambda x: M._lin(x, sd, p + ".to_q"), to_k=lambda x: M._lin(x, sd, p + ".to_k"),
                    to_v=lambda x: M._lin(x, sd, p + ".to_v"), to_out=lambda x: M._lin(x, sd, p + ".to_out.0"),
                    reshape_heads_to_batch_dim=to_batch, reshape_batch_dim_to_heads=to_heads, _slice_size=None, heads=heads,
                    scale=d ** -0.5, _tome_info={"r": [0] * 64, "class_token": False, "distill_token": False, "trace_source": False})

    def ref_attn(p, heads):
        ca = cross_attention(p, heads)
        return lambda x, context=None: T.ToMeCrossAttention.forward(ca, x, context=context)

    # ---- BasicTransformerBlock (tome_unet.py:129-136) ------------------------------------------------------------------
    def ref_tblock(p, heads):
        C = sd[p + ".norm1.weight"].shape[0]
        ln = lambda n: (lambda x: F.layer_norm(x, (C,), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1e-5))
        a1, a2 = ref_attn(p + ".attn1", heads), ref_attn(p + ".attn2", heads)

        def ff(x):
            h = M._lin(x, sd, p + ".ff.net.0.proj")
            val, gate = h.chunk(2, dim=-1)
            return M._lin(val * F.gelu(gate), sd, p + ".ff.net.2")
        blk = make(T.ToMeTransformerBlock, norm1=ln("norm1"), norm2=ln("norm2"), norm3=ln("norm3"),
                   attn1=lambda x, size=None: (a1(x), None), attn2=lambda x, context=None: a2(x, context=context), ff=ff)
        return lambda x, context=None: T.ToMeTransformerBlock.forward(blk, x, context=context)

    # ---- SpatialTransformer (tome_unet.py:114-127) ---------------------------------------------------------------------
    def ref_spatial(p, heads, depth):
        st = make(T.ToMeSpatialTransformer, norm=lambda x: M._gn(x, sd, p + ".norm", g, 1e-6),
                  proj_in=lambda x: M._conv(x, sd, p + ".proj_in", padding=0),
                  proj_out=lambda x: M._conv(x, sd, p + ".proj_out", padding=0),
                  transformer_blocks=[ref_tblock(f"{p}.transformer_blocks.{d}", heads) for d in range(depth)])
        # the ToMe variants of the blocks call attn(hidden, context=...) and expect (hidden, context) back
        return lambda x, context=None: (T.ToMeSpatialTransformer.forward(st, x, context=context), context)

    x = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([981, 17])
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=torch.Generator().manual_seed(2))
    taps = {}
    ref_out = M.unet_forward(sd, cfg, x, t, ctx, taps=taps)

    # leaf-level comparisons first
    p = "down_blocks.0.attentions.0"
    h0 = M.resnet_block(M._conv(x, sd, "conv_in"), taps["temb"], sd, "down_blocks.0.resnets.0", g, eps)
    got, _ = ref_spatial(p, cfg.num_heads[0], 1)(h0, context=ctx)
    want = M.transformer_2d(h0, ctx, sd, p, cfg.num_heads[0], g, 1, False)
    out["spatial_transformer_max_abs"] = float((got - want).abs().max())
    tok = torch.randn(2, 64, boc[0], generator=torch.Generator().manual_seed(3))
    got = ref_tblock(p + ".transformer_blocks.0", cfg.num_heads[0])(tok, context=ctx)
    want = M.basic_transformer_block(tok, ctx, sd, p + ".transformer_blocks.0", cfg.num_heads[0])
    out["transformer_block_max_abs"] = float((got - want).abs().max())
    got = ref_attn(p + ".transformer_blocks.0.attn2", cfg.num_heads[0])(tok, context=ctx)
    pp = p + ".transformer_blocks.0.attn2"
    want = M._lin(M.attention(M._lin(tok, sd, pp + ".to_q"), M._lin(ctx, sd, pp + ".to_k"), M._lin(ctx, sd, pp + ".to_v"),
                              cfg.num_heads[0]), sd, pp + ".to_out.0")
    out["cross_attention_max_abs"] = float((got - want).abs().max())

    # ---- the trunk through the reference's block forwards ---------------------------------------------------------------
    temb = taps["temb"]
    res = lambda p_: (lambda h, te=None: M.resnet_block(h, temb, sd, p_, g, eps))
    n = len(boc)
    h = M._conv(x, sd, "conv_in")
    skips = (h,)
    level_err = {}
    for i in range(n):
        pre = f"down_blocks.{i}"
        down = None if i == n - 1 else [lambda hh, i=i: M._conv(hh, sd, f"down_blocks.{i}.downsamplers.0.conv", stride=2, padding=1)]
        resnets = [res(f"{pre}.resnets.{j}") for j in range(cfg.layers_per_block)]
        if cfg.attn_levels[i]:
            blk = make(T.ToMeDownBlock, resnets=resnets, downsamplers=down, training=False, gradient_checkpointing=False,
                       attentions=[ref_spatial(f"{pre}.attentions.{j}", cfg.num_heads[i], cfg.transformer_depth[i])
                                   for j in range(cfg.layers_per_block)])
            h, states, _ = T.ToMeDownBlock.forward(blk, h, temb, ctx)
        else:                                   # DownBlock2D has no vendored copy: same loop without the attentions
            states = ()
            for r_ in resnets:
                h = r_(h, temb)
                states += (h,)
            if down is not None:
                h = down[0](h)
                states += (h,)
        skips += states
        level_err[f"down{i}"] = float((h - taps[f"down{i}"]).abs().max())
    mid = make(T.ToMeMidBlock, resnets=[res("mid_block.resnets.0"), res("mid_block.resnets.1")],
               attentions=[ref_spatial("mid_block.attentions.0", cfg.num_heads[-1], cfg.transformer_depth[-1])])
    h, _ = T.ToMeMidBlock.forward(mid, h, temb, ctx)
    level_err["mid"] = float((h - taps["mid"]).abs().max())
    for i in range(n):
        lvl = n - 1 - i
        pre = f"up_blocks.{i}"
        k = cfg.layers_per_block + 1
        res_tuple, skips = skips[-k:], skips[:-k]
        ups = None
        if i < n - 1:
            tgt = skips[-1].shape[-2:]
            ups = [lambda hh, size=None, i=i, tgt=tgt: M._conv(F.interpolate(hh, size=tgt, mode="nearest"), sd, f"up_blocks.{i}.upsamplers.0.conv")]
        resnets = [res(f"{pre}.resnets.{j}") for j in range(k)]
        if cfg.attn_levels[lvl]:
            blk = make(T.ToMeUpBlock, resnets=resnets, upsamplers=ups, training=False, gradient_checkpointing=False,
                       attentions=[ref_spatial(f"{pre}.attentions.{j}", cfg.num_heads[lvl], cfg.transformer_depth[lvl]) for j in range(k)])
            h, _ = T.ToMeUpBlock.forward(blk, h, res_tuple, temb, ctx)
        else:                                   # UpBlock2D: the vendored CrossAttnUpBlock2D loop without the attentions
            for r_ in resnets:
                h = r_(torch.cat([h, res_tuple[-1]], dim=1), temb)
                res_tuple = res_tuple[:-1]
            if ups is not None:
                h = ups[0](h)
        level_err[f"up{i}"] = float((h - taps[f"up{i}"]).abs().max())
    h = M._conv(F.silu(M._gn(h, sd, "conv_norm_out", g, eps)), sd, "conv_out")
    out["levels_max_abs"] = level_err
    out["unet_out_max_abs"] = float((h - ref_out).abs().max())
    out["unet_out_absmax"] = float(ref_out.abs().max())
    out["skips_consumed"] = len(skips) == 0
    print("PROBE_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
