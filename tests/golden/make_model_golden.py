"""Pin oracle/models_ref.py against the REAL third-party modules, where they exist.

The UNet / VAE arithmetic of the reference lives in diffusers ~= 0.16.0 (reference pyproject.toml:22), which is neither
vendored in /root/reference nor installed in the build container, so oracle/models_ref.py is "parity unpinned"
(SURVEY.md 8c).  This script closes that gap on any machine that has diffusers:

    pip install "diffusers~=0.16.0"        # (not possible in the build container: no network)
    python tests/golden/make_model_golden.py

It instantiates diffusers.UNet2DConditionModel / AutoencoderKL with the tiny and the SD1.x hyper-parameters, loads the
SAME seeded synthetic weights the tests use (gyre_amd.weights.synthetic_state_dict - the key space is diffusers' own),
runs them in fp32 on the CPU and writes inputs + outputs to tests/golden/model_vectors.npz (small: tensors of the tiny
model and a strided sample of the SD1.x outputs).  tests/test_oracle_model_golden.py then checks the oracle against that
file; while the file is absent that test is skipped and the oracle stays unpinned.

If SD_WEIGHT_ROOT points at a diffusers-layout SD1.5 folder the real checkpoint is used for the full-size case instead of
synthetic weights (only the output sample is stored, never weights).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    try:
        import diffusers
        from diffusers import AutoencoderKL, UNet2DConditionModel
    except Exception as e:  # noqa: BLE001
        print(f"diffusers is not importable here ({e}); nothing written - oracle/models_ref.py stays PARITY UNPINNED.")
        return 2
    from gyre_amd import config as gcfg, weights
    out = {"diffusers_version": np.array(diffusers.__version__)}
    g = torch.Generator().manual_seed(1234)

    def unet_of(cfg):
        n = len(cfg.block_out_channels)
        return UNet2DConditionModel(
            in_channels=cfg.in_channels, out_channels=cfg.out_channels, block_out_channels=cfg.block_out_channels,
            layers_per_block=cfg.layers_per_block, cross_attention_dim=cfg.cross_attention_dim,
            attention_head_dim=cfg.num_heads, norm_num_groups=cfg.norm_num_groups, sample_size=cfg.sample_size,
            down_block_types=tuple("CrossAttnDownBlock2D" if a else "DownBlock2D" for a in cfg.attn_levels),
            up_block_types=tuple("CrossAttnUpBlock2D" if a else "UpBlock2D" for a in reversed(cfg.attn_levels)),
            flip_sin_to_cos=cfg.flip_sin_to_cos, freq_shift=int(cfg.freq_shift), use_linear_projection=cfg.use_linear_projection)

    def vae_of(cfg):
        n = len(cfg.block_out_channels)
        return AutoencoderKL(in_channels=cfg.in_channels, out_channels=cfg.out_channels, latent_channels=cfg.latent_channels,
                             block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
                             norm_num_groups=cfg.norm_num_groups, down_block_types=("DownEncoderBlock2D",) * n,
                             up_block_types=("UpDecoderBlock2D",) * n, sample_size=cfg.sample_size)

    with torch.no_grad():
        for name, ucfg, lat, full in (("tiny", gcfg.tiny_unet(), 16, True), ("tiny9", gcfg.tiny_unet(in_channels=9), 16, True),
                                      ("sd15", gcfg.sd15_unet(), 32, False)):
            net = unet_of(ucfg).eval()
            sd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 7)
            root = os.environ.get("SD_WEIGHT_ROOT")
            if name == "sd15" and root:
                from safetensors.torch import load_file
                sd = load_file(os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))
                out["sd15_real_checkpoint"] = np.array(1)
            missing, unexpected = net.load_state_dict(sd, strict=False)
            assert not missing and not unexpected, (missing[:3], unexpected[:3])
            x = torch.randn(2, ucfg.in_channels, lat, lat + 8, generator=g)
            t = torch.tensor([981, 17])
            ctx = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g)
            y = net(x, t, encoder_hidden_states=ctx).sample
            out[f"unet_{name}_x"], out[f"unet_{name}_t"], out[f"unet_{name}_ctx"] = x.numpy(), t.numpy(), ctx.numpy()
            out[f"unet_{name}_eps"] = y.numpy() if full else y[:, :, ::3, ::5].numpy()
        for name, vcfg, px, full in (("tiny", gcfg.tiny_vae(), 64, True), ("sd15", gcfg.sd15_vae(), 128, False)):
            vae = vae_of(vcfg).eval()
            sd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 8)
            ren = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}
            have = set(vae.state_dict())
            sd2 = {}
            for k, v in sd.items():
                if k not in have:
                    for a, b in ren.items():
                        k = k.replace(a, b)
                sd2[k] = v
            vae.load_state_dict(sd2)
            img = torch.rand(1, 3, px, px + 32, generator=g) * 2 - 1
            mom = vae.encode(img).latent_dist.parameters
            z = torch.randn(1, vcfg.latent_channels, px // 8, px // 8 + 4, generator=g)
            dec = vae.decode(z).sample
            out[f"vae_{name}_img"], out[f"vae_{name}_z"] = img.numpy(), z.numpy()
            out[f"vae_{name}_moments"] = mom.numpy()
            out[f"vae_{name}_dec"] = dec.numpy() if full else dec[:, :, ::7, ::5].numpy()
    path = os.path.join(HERE, "model_vectors.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB) with diffusers {diffusers.__version__}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
