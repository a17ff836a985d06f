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

    class UpBlock2D:                           # UpBlock2D has no vendored copy: the vendored CrossAttnUpBlock2D loop without attentions
        def __init__(self, resnets, ups):
            self.resnets, self.upsamplers = resnets, ups

        def __call__(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None):
            for r_ in self.resnets:
                hidden_states = r_(torch.cat([hidden_states, res_hidden_states_tuple[-1]], dim=1), temb)
                res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            if self.upsamplers is not None:
                hidden_states = self.upsamplers[0](hidden_states)
            return hidden_states

    class Callable(torch.nn.Module):
        def __init__(self, fn, resnets=None):
            super().__init__()
            self.fn = fn
            if resnets is not None:
                self.resnets = resnets

        def forward(self, *a, **k):
            return self.fn(*a, **k)

    def trunk(down_res=None, mid_res=None, want_taps=None, adapters=None):
        if adapters is not None:
            from gyre.pipeline.t2i_adapter import unet_patcher as TP      # the reference's OWN T2I-adapter patcher
        h = M._conv(x, sd, "conv_in")
        skips = (h,)
        level_err = {}
        for i in range(n):
            pre = f"down_blocks.{i}"
            down = None if i == n - 1 else torch.nn.ModuleList(
                [Callable(lambda hh, i=i: M._conv(hh, sd, f"down_blocks.{i}.downsamplers.0.conv", stride=2, padding=1))])
            resnets = [res(f"{pre}.resnets.{j}") for j in range(cfg.layers_per_block)]
            if cfg.attn_levels[i]:
                blk = make(T.ToMeDownBlock, resnets=resnets, downsamplers=down, training=False, gradient_checkpointing=False,
                           attentions=[ref_spatial(f"{pre}.attentions.{j}", cfg.num_heads[i], cfg.transformer_depth[i])
                                       for j in range(cfg.layers_per_block)])
                if adapters is None:
                    h, states, _ = T.ToMeDownBlock.forward(blk, h, temb, ctx)
                else:
                    # what DownBlockWrapper.forward does for a cross-attention block: hand adapter_state to the block, whose
                    # CrossAttnDownBlock2DHook (t2i_adapter/unet_patcher.py:33-61) wraps the downsampler / adds afterwards
                    hook = TP.CrossAttnDownBlock2DHook()
                    args, kwargs = hook.pre_forward(blk, h, temb, ctx, adapter_state=adapters[i])
                    hh, states, _ = T.ToMeDownBlock.forward(blk, *args, **kwargs)
                    h, states = hook.post_forward(blk, (hh, states))
            else:                               # DownBlock2D has no vendored copy: same loop without the attentions
                def plain_block(hh, te=None, **_):
                    st_ = ()
                    for r_ in resnets:
                        hh = r_(hh, te)
                        st_ += (hh,)
                    if down is not None:
                        hh = down[0](hh)
                        st_ += (hh,)
                    return hh, st_
                if adapters is None:
                    h, states = plain_block(h, temb)
                else:
                    h, states = TP.DownBlockWrapper(Callable(plain_block), adapters[i])(h, temb)
            skips += states
            if want_taps is not None:
                level_err[f"down{i}"] = float((h - want_taps[f"down{i}"]).abs().max())
        mid_obj = make(T.ToMeMidBlock, resnets=[res("mid_block.resnets.0"), res("mid_block.resnets.1")],
                       attentions=[ref_spatial("mid_block.attentions.0", cfg.num_heads[-1], cfg.transformer_depth[-1])])
        mid_block = lambda hh, te, c_: T.ToMeMidBlock.forward(mid_obj, hh, te, c_)[0]
        up_blocks = []
        sizes = [s_.shape[-2:] for s_ in skips]
        consumed = len(skips)
        for i in range(n):
            lvl = n - 1 - i
            pre = f"up_blocks.{i}"
            k = cfg.layers_per_block + 1
            consumed -= k
            ups = None
            if i < n - 1:
                tgt = sizes[consumed - 1]
                ups = [lambda hh, size=None, i=i, tgt=tgt: M._conv(F.interpolate(hh, size=tgt, mode="nearest"), sd, f"up_blocks.{i}.upsamplers.0.conv")]
            resnets = [res(f"{pre}.resnets.{j}") for j in range(k)]
            if cfg.attn_levels[lvl]:
                blk = make(T.ToMeUpBlock, resnets=resnets, upsamplers=ups, training=False, gradient_checkpointing=False,
                           attentions=[ref_spatial(f"{pre}.attentions.{j}", cfg.num_heads[lvl], cfg.transformer_depth[lvl]) for j in range(k)])
                up_blocks.append((lambda blk: (lambda hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None:
                                               T.ToMeUpBlock.forward(blk, hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states)[0]))(blk))
                up_blocks[-1].resnets = resnets
            else:
                up_blocks.append(UpBlock2D(resnets, ups))
        if down_res is not None or mid_res is not None:
            # the reference's OWN ControlNet patcher (gyre/pipeline/controlnet/unet_patcher.py:60-95) rewires the up / mid blocks
            from gyre.pipeline.controlnet import unet_patcher as P

            plain_mid = mid_block
            module = SimpleNamespace(up_blocks=[Callable(b, b.resnets) for b in up_blocks], mid_block=Callable(plain_mid))
            hook = P.UNet2DConditionModelHook()
            hook.pre_forward(module, down_block_additional_residuals=down_res, mid_block_additional_residual=mid_res)
            up_blocks, mid_block = list(module.up_blocks), module.mid_block
        h = mid_block(h, temb, ctx)
        if want_taps is not None:
            level_err["mid"] = float((h - want_taps["mid"]).abs().max())
        for i in range(n):
            k = cfg.layers_per_block + 1
            res_tuple, skips = skips[-k:], skips[:-k]
            h = up_blocks[i](hidden_states=h, res_hidden_states_tuple=res_tuple, temb=temb, encoder_hidden_states=ctx)
            if want_taps is not None:
                level_err[f"up{i}"] = float((h - want_taps[f"up{i}"]).abs().max())
        h = M._conv(F.silu(M._gn(h, sd, "conv_norm_out", g, eps)), sd, "conv_out")
        return h, level_err, len(skips) == 0

    h, level_err, consumed_all = trunk(want_taps=taps)
    out["levels_max_abs"] = level_err
    out["unet_out_max_abs"] = float((h - ref_out).abs().max())
    out["unet_out_absmax"] = float(ref_out.abs().max())
    out["skips_consumed"] = consumed_all

    # ---- ControlNet residual injection: the reference's patcher vs the oracle's down_res / mid_res -------------------------
    shapes, hh, ww = [(boc[0], 16, 16)], 16, 16
    for i, c_ in enumerate(boc):
        shapes += [(c_, hh, ww)] * cfg.layers_per_block
        if i < n - 1:
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
            shapes.append((c_, hh, ww))
    gen = torch.Generator().manual_seed(11)
    down_res = [torch.randn(2, *s_, generator=gen) * 0.5 for s_ in shapes]
    mid_res = torch.randn(2, boc[-1], hh, ww, generator=gen) * 0.5
    want = M.unet_forward(sd, cfg, x, t, ctx, down_res=down_res, mid_res=mid_res)
    got, _, _ = trunk(down_res=list(down_res), mid_res=mid_res)
    out["controlnet_patcher_max_abs"] = float((got - want).abs().max())
    out["controlnet_effect"] = float((want - ref_out).abs().max())

    # ---- T2I-adapter states: the reference's patcher (in-place adds inside the down path) vs the oracle's adapter_states -----
    ashapes, hh, ww = [], 16, 16
    for i, c_ in enumerate(boc):
        ashapes.append((c_, hh, ww))
        if i < n - 1:
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
    adapters = [torch.randn(2, *s_, generator=gen) * 0.5 for s_ in ashapes]
    want = M.unet_forward(sd, cfg, x, t, ctx, adapter_states=adapters)
    got, _, _ = trunk(adapters=[a.clone() for a in adapters])
    out["t2i_patcher_max_abs"] = float((got - want).abs().max())
    out["t2i_effect"] = float((want - ref_out).abs().max())
    print("PROBE_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
