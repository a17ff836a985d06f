"""Model hyper-parameters for the native UNet / VAE.

The SD1.x constants are the ones the reference pins in-tree at
``gyre/ldm_config/v1-inference.yaml:29-64`` (UNet: model_channels 320,
channel_mult [1,2,4,4], 2 res blocks, attention at the first three levels,
8 heads, context_dim 768; VAE: ch 128, ch_mult [1,2,4,4], z=4, double_z).
Field names follow the diffusers ``config.json`` keys the reference reads
(``unified_pipeline.py:186`` in_channels, ``:1318`` sample_size,
``:1402`` block_out_channels).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    attn_levels: Tuple[bool, ...] = (True, True, True, False)
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    transformer_depth: Tuple[int, ...] = (1, 1, 1, 1)
    use_linear_projection: bool = False
    sample_size: int = 64
    flip_sin_to_cos: bool = True
    freq_shift: float = 0.0
    # SDXL "text_time" added conditioning (None for SD1.x/2.x)
    addition_embed_type: str | None = None
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    _diffusers_version: str = "0.16.0"

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    def to_dict(self):
        return asdict(self)


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215
    sample_size: int = 512

    def to_dict(self):
        return asdict(self)


def sd15_unet(in_channels: int = 4) -> UNetConfig:
    """SD1.x UNet (in_channels=9 for the runway inpainting variant,
    reference unified_pipeline.py:648-696)."""
    return UNetConfig(in_channels=in_channels)


def sd15_vae() -> VAEConfig:
    return VAEConfig()


def tiny_unet(in_channels: int = 4) -> UNetConfig:
    """Same topology as SD1.x at 1/10 width - for parity tests that must run in
    seconds on the CPU oracle.  Channels stay multiples of 32 (GroupNorm groups)
    and head dims multiples of 8 (vector loads)."""
    return UNetConfig(in_channels=in_channels, block_out_channels=(32, 64, 128, 128),
                      num_heads=(2, 2, 4, 4), cross_attention_dim=64, sample_size=16)


def sdxl_unet() -> UNetConfig:
    """SDXL-base UNet (BASELINE config 4; not in the reference - an extension on the same kernel set): 3 levels,
    transformer depth 0/2/10, head dim 64, context 2048, linear projections, text_time added conditioning."""
    return UNetConfig(block_out_channels=(320, 640, 1280), attn_levels=(False, True, True), num_heads=(5, 10, 20),
                      transformer_depth=(1, 2, 10), cross_attention_dim=2048, use_linear_projection=True,
                      sample_size=128, addition_embed_type="text_time")


def tiny_sdxl_unet() -> UNetConfig:
    return UNetConfig(block_out_channels=(32, 64, 128), attn_levels=(False, True, True), num_heads=(2, 2, 4),
                      transformer_depth=(1, 2, 3), cross_attention_dim=64, use_linear_projection=True, sample_size=16,
                      addition_embed_type="text_time", addition_time_embed_dim=8,
                      projection_class_embeddings_input_dim=32 + 6 * 8)


def sdxl_vae() -> VAEConfig:
    """SDXL's AutoencoderKL: same topology as SD1.x, latent scaling 0.13025, 1024 px sample size."""
    return VAEConfig(scaling_factor=0.13025, sample_size=1024)


def tiny_vae() -> VAEConfig:
    return VAEConfig(block_out_channels=(32, 64, 64, 64), sample_size=64)
