"""In-tree witnesses for the UNet/VAE oracle (build container only: needs /root/reference).

oracle/models_ref.py restates diffusers ~= 0.16's UNet2DConditionModel / AutoencoderKL, which is neither vendored nor
installed, and the reference's tests hold no tensors: its parity stays UNPINNED (oracle header).  What the
reference tree DOES hold are three independent statements of the same architecture; every hyper-parameter and piece of
wiring they expose is checked here against the oracle and the product config:

  * gyre/pipeline/controlnet/models.py:96-279  ControlNetModel - a vendored copy of the UNet's ENCODER half: constructor
    defaults (flip_sin_to_cos, freq_shift, downsample_padding, act_fn, norm_eps, groups, channels, block types,
    "attention_head_dim" = number of heads) and how they are passed to the blocks (read with `ast`, nothing is imported)
  * gyre/pipeline/controlnet/unet_patcher.py:64-84  how the decoder consumes the skip connections (len(resnets) per up block,
    from the END of the tuple)
  * gyre/ldm_config/v1-inference.yaml:1-70  the SD1.x UNet / VAE / schedule constants
"""
import ast
import os
import re

import pytest
import torch

from gyre_amd import config as gcfg, weights
from oracle import models_ref as M

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")


def _ctor(path, cls):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == "__init__":
                    return fn
    raise AssertionError(f"{cls}.__init__ not found in {path}")


def _defaults(fn):
    args = fn.args.args[1:]
    vals = fn.args.defaults
    out = {}
    for a, d in zip(args[len(args) - len(vals):], vals):
        try:
            out[a.arg] = ast.literal_eval(d)
        except ValueError:
            out[a.arg] = ast.unparse(d)
    return out


def _call_kwargs(fn, callee):
    """keyword arguments (as source text) of every call to `callee` inside fn"""
    out = []
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and getattr(node.func, "id", getattr(node.func, "attr", None)) == callee:
            out.append({k.arg: ast.unparse(k.value) for k in node.keywords})
    return out


def test_controlnet_encoder_defaults_match_oracle_and_product_config():
    fn = _ctor("gyre/pipeline/controlnet/models.py", "ControlNetModel")
    d = _defaults(fn)
    ucfg, ocfg = gcfg.sd15_unet(), M.UNetRefConfig()
    for cfg in (ucfg, ocfg):
        assert d["in_channels"] == cfg.in_channels == 4
        assert d["flip_sin_to_cos"] is True and cfg.flip_sin_to_cos is True          # [cos, sin] order
        assert d["freq_shift"] == cfg.freq_shift == 0
        assert tuple(d["block_out_channels"]) == tuple(cfg.block_out_channels) == (320, 640, 1280, 1280)
        assert d["layers_per_block"] == cfg.layers_per_block == 2
        assert d["norm_num_groups"] == cfg.norm_num_groups == 32
        assert tuple("CrossAttn" in t for t in d["down_block_types"]) == tuple(cfg.attn_levels)
        assert d["use_linear_projection"] is False and cfg.use_linear_projection is False
        # SD1.x "attention_head_dim: 8" is handed on as attn_num_head_channels = NUMBER of heads per level
        assert d["attention_head_dim"] == 8 and tuple(cfg.num_heads) == (8, 8, 8, 8)
    assert d["act_fn"] == "silu" and d["downsample_padding"] == 1 and d["norm_eps"] == 1e-5
    assert d["resnet_time_scale_shift"] == "default" and d["class_embed_type"] is None and d["upcast_attention"] is False


def test_controlnet_encoder_wiring_matches_oracle():
    fn = _ctor("gyre/pipeline/controlnet/models.py", "ControlNetModel")
    src = ast.unparse(fn)
    # time embedding: sinusoid of width block_out_channels[0] -> MLP to 4x that width
    assert "time_embed_dim = block_out_channels[0] * 4" in src
    assert re.search(r"Timesteps\(block_out_channels\[0\], flip_sin_to_cos, freq_shift\)", src)
    assert re.search(r"TimestepEmbedding\(\s*timestep_input_dim, time_embed_dim", src)
    assert gcfg.sd15_unet().time_embed_dim == 1280
    # conv_in: 3x3, padding 1
    assert "conv_in_kernel = 3" in src and "conv_in_padding = (conv_in_kernel - 1) // 2" in src
    assert re.search(r"self\.conv_in = nn\.Conv2d\(in_channels, block_out_channels\[0\], kernel_size=conv_in_kernel, padding=conv_in_padding\)", src)
    (down,) = _call_kwargs(fn, "get_down_block")
    assert down["resnet_eps"] == "norm_eps" and down["resnet_act_fn"] == "act_fn" and down["resnet_groups"] == "norm_num_groups"
    assert down["attn_num_head_channels"] == "attention_head_dim[i]"                  # heads, per level
    assert down["downsample_padding"] == "downsample_padding" and down["add_downsample"] == "not is_final_block"
    assert down["temb_channels"] == "time_embed_dim" and down["num_layers"] == "layers_per_block"
    (mid,) = _call_kwargs(fn, "UNetMidBlock2DCrossAttn")
    assert mid["resnet_eps"] == "norm_eps" and mid["attn_num_head_channels"] == "attention_head_dim[-1]"
    # the oracle's constants for exactly these knobs
    osrc = open(M.__file__).read()
    assert "g, eps = cfg.norm_num_groups, 1e-5" in osrc                                # ResnetBlock2D GroupNorm eps
    assert 'f"down_blocks.{i}.downsamplers.0.conv", stride=2, padding=1' in osrc       # downsample_padding 1
    assert "F.silu(" in osrc
    # forward order of the encoder half: conv_in -> per level [resnet (+ attention)] * layers, downsample except last -> mid
    fwd = None
    for node in ast.walk(ast.parse(open(os.path.join(REF, "gyre/pipeline/controlnet/models.py")).read())):
        if isinstance(node, ast.ClassDef) and node.name == "ControlNetModel":
            fwd = next(f for f in node.body if isinstance(f, ast.FunctionDef) and f.name == "forward")
    fsrc = ast.unparse(fwd)
    order = [fsrc.index(s) for s in ("self.time_proj(", "self.time_embedding(", "self.conv_in(", "for downsample_block in self.down_blocks",
                                     "self.mid_block(")]
    assert order == sorted(order)
    assert "down_block_res_samples = (sample,)" in fsrc                                # conv_in's output is the first skip


def test_skip_connection_bookkeeping_matches_unet_patcher():
    """unet_patcher.py:64-84 hands each up block len(block.resnets) residuals from the END of the skip tuple; the SD1.x
    tuple has 12 entries (conv_in + per level: layers + downsampler).  Same counts from the product's parameter table
    and the oracle's loop structure (which pops one skip per up-block resnet)."""
    src = open(os.path.join(REF, "gyre/pipeline/controlnet/unet_patcher.py")).read()
    assert "chunk_size = len(block.resnets)" in src and "down_residuals[-chunk_size:]" in src
    cfg = gcfg.sd15_unet()
    shapes = weights.unet_param_shapes(cfg)
    n = len(cfg.block_out_channels)
    skips = 1 + sum(cfg.layers_per_block + (1 if i < n - 1 else 0) for i in range(n))
    up_resnets = [len({k.split(".")[3] for k in shapes if k.startswith(f"up_blocks.{i}.resnets.")}) for i in range(n)]
    assert skips == 12 and up_resnets == [3, 3, 3, 3] and sum(up_resnets) == skips
    # the oracle consumes exactly that many: a forward on the tiny config leaves no skip behind and needs none extra
    tcfg = gcfg.tiny_unet()
    sd = weights.synthetic_state_dict(weights.unet_param_shapes(tcfg))
    out = M.unet_forward(sd, tcfg, torch.zeros(1, 4, 16, 16), torch.tensor([1]), torch.zeros(1, 77, tcfg.cross_attention_dim))
    assert out.shape == (1, 4, 16, 16)
    # concat widths of the first resnet of every up block = own channels + popped skip channels
    boc = cfg.block_out_channels
    assert shapes["up_blocks.0.resnets.0.conv1.weight"][1] == boc[3] + boc[3]
    assert shapes["up_blocks.1.resnets.2.conv1.weight"][1] == boc[2] + boc[1]
    assert shapes["up_blocks.3.resnets.2.conv1.weight"][1] == boc[0] + boc[0]


def test_v1_inference_yaml_matches_configs_and_schedule():
    import yaml
    y = yaml.safe_load(open(os.path.join(REF, "gyre/ldm_config/v1-inference.yaml")))["model"]["params"]
    u = y["unet_config"]["params"]
    ucfg = gcfg.sd15_unet()
    assert u["in_channels"] == ucfg.in_channels and u["out_channels"] == ucfg.out_channels
    assert tuple(u["model_channels"] * m for m in u["channel_mult"]) == tuple(ucfg.block_out_channels)
    assert u["num_res_blocks"] == ucfg.layers_per_block and u["num_heads"] == ucfg.num_heads[0]
    assert u["context_dim"] == ucfg.cross_attention_dim and u["transformer_depth"] == ucfg.transformer_depth[0]
    # attention at downsample rates 1, 2, 4 = the first three levels
    assert sorted(u["attention_resolutions"]) == [1, 2, 4] and tuple(ucfg.attn_levels) == (True, True, True, False)
    v = y["first_stage_config"]["params"]["ddconfig"]
    vcfg = gcfg.sd15_vae()
    assert tuple(v["ch"] * m for m in v["ch_mult"]) == tuple(vcfg.block_out_channels)
    assert v["z_channels"] == vcfg.latent_channels and v["double_z"] is True and v["num_res_blocks"] == vcfg.layers_per_block
    assert v["in_channels"] == vcfg.in_channels and v["out_ch"] == vcfg.out_channels and v["attn_resolutions"] == []
    assert y["scale_factor"] == vcfg.scaling_factor == 0.18215
    from gyre_amd import schedulers as S
    sch = S.DiscreteSchedule(y["timesteps"])
    betas = torch.linspace(y["linear_start"] ** 0.5, y["linear_end"] ** 0.5, y["timesteps"], dtype=torch.float32) ** 2
    ac = torch.cumprod(1 - betas, 0)
    assert torch.allclose(sch.alphas_cumprod.float(), ac, rtol=1e-6)
    # parameter counts that follow from those constants (published SD1.x figures)
    assert sum(int(torch.tensor(s).prod()) for s in weights.unet_param_shapes(ucfg).values()) == 859_520_964
