"""CPU-only checks of the host side: library exports, module trees, error behaviour."""
import os
import re
import subprocess

import pytest
import torch

from gyre_amd import _lib, config as gcfg, weights
from gyre_amd.modules import GyreHipUNet, GyreHipVAE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gyre_hip.h")).read()
    declared = set(re.findall(r"\b(gyre_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for storage in (_lib.BF16, _lib.F16):           # both storage builds export the same ABI
        L = _lib.lib(storage)  # loads and binds every symbol of _SIGS
        for name in declared:
            assert hasattr(L, name), f"{name} declared in gyre_hip.h but not exported"
        assert L.gyre_abi_version() == 1 and L.gyre_storage_dtype() == storage
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)


def test_unet_state_dict_is_diffusers_keyed():
    net = GyreHipUNet(gcfg.tiny_unet())
    keys = list(net.state_dict().keys())
    assert sorted(keys) == sorted(weights.unet_param_shapes(gcfg.tiny_unet()).keys())
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight" in keys
    assert "up_blocks.3.resnets.2.conv_shortcut.weight" in keys
    assert net.config.in_channels == 4 and net.config.sample_size == 16
    assert sum(1 for _ in net.modules()) > 100  # LoRA sweep walks .modules()


def test_sd15_param_counts():
    n = sum(torch.Size(s).numel() for s in weights.unet_param_shapes(gcfg.sd15_unet()).values())
    assert n == 859_520_964  # SD1.x UNet
    n = sum(torch.Size(s).numel() for s in weights.vae_param_shapes(gcfg.sd15_vae()).values())
    assert n == 83_653_863   # SD1.x AutoencoderKL


def test_vae_accepts_new_attention_key_names():
    cfg = gcfg.tiny_vae()
    sd = weights.synthetic_state_dict(weights.vae_param_shapes(cfg))
    ren = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}
    new = {}
    for k, v in sd.items():
        for a, b in ren.items():
            k = k.replace(a, b)
        new[k] = v
    net = GyreHipVAE(cfg)
    net.load_state_dict(new)
    assert torch.equal(net.state_dict()["decoder.mid_block.attentions.0.query.weight"],
                       sd["decoder.mid_block.attentions.0.query.weight"])


def test_no_cpu_fallback():
    net = GyreHipUNet(gcfg.tiny_unet())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 4, 16, 16), 1, encoder_hidden_states=torch.zeros(1, 77, 64))


def test_product_does_not_import_oracle():
    out = subprocess.run(["grep", "-rlE", r"^\s*(from|import)\s+oracle", os.path.join(ROOT, "gyre_amd")],
                         capture_output=True, text=True).stdout.strip()
    assert out == "", f"product code imports the oracle: {out}"


def test_sdxl_param_count_and_keys():
    shapes = weights.unet_param_shapes(gcfg.sdxl_unet())
    assert sum(torch.Size(s).numel() for s in shapes.values()) == 2_567_463_684  # SDXL-base UNet
    assert "down_blocks.2.attentions.1.transformer_blocks.9.attn2.to_k.weight" in shapes
    assert shapes["down_blocks.1.attentions.0.proj_in.weight"] == (640, 640)        # linear projection
    assert shapes["add_embedding.linear_1.weight"] == (1280, 2816)
    assert not any(k.startswith("down_blocks.0.attentions") for k in shapes)


def test_counted_vmcnt_kernels_do_not_spill():
    """Kernels that wait for LDS-DMA tiles with a counted s_waitcnt vmcnt(N) must have no scratch traffic (spill
    stores also count in vmcnt and may retire before older loads).  gyre_amd/build.py records clang's per-kernel
    resource remarks at compile time; only the allow-listed kernel (which drains with vmcnt(0)) may use scratch."""
    import json
    from gyre_amd import build as B
    B.build()
    data = json.load(open(B.RESOURCES_JSON))
    assert set(data) == set(B.RESOURCE_FILES) | {f"f16/{f}" for f in B.RESOURCE_FILES}      # both storage flavours
    n = 0
    for src, kernels in data.items():
        for name, r in kernels.items():
            n += 1
            if r.get("scratch_bytes_per_lane", 0) > 0:
                assert any(a in name for a in B.SCRATCH_ALLOWED), f"{name} in {src} spills {r['scratch_bytes_per_lane']} B/lane"
    assert n > 80


def test_from_pretrained_reads_diffusers_layout(tmp_path):
    """config.json + safetensors in the HF diffusers folder layout, without diffusers (reference manager.py:1176-1252
    calls Class.from_pretrained(path, torch_dtype=..., variant=...)): SD1.x-style and SDXL-style UNet configs, variant
    selection, the >= 0.17 VAE attention key names, and the pipeline-level loader."""
    import json
    from safetensors.torch import save_file
    from gyre_amd import config as gcfg, weights
    from gyre_amd.modules import GyreHipUNet, GyreHipVAE
    from gyre_amd.pipeline import GyrePipeline
    root = tmp_path / "model"
    (root / "unet").mkdir(parents=True)
    (root / "vae").mkdir()
    (root / "model_index.json").write_text(json.dumps({"_class_name": "StableDiffusionPipeline"}))
    ucfg = gcfg.tiny_unet()
    (root / "unet" / "config.json").write_text(json.dumps({
        "_class_name": "UNet2DConditionModel", "in_channels": 4, "out_channels": 4, "sample_size": 16,
        "block_out_channels": [32, 64, 128, 128], "layers_per_block": 2, "cross_attention_dim": 64,
        "attention_head_dim": [2, 2, 4, 4], "norm_num_groups": 32,
        "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"]}))
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
    save_file({k: v.contiguous() for k, v in usd.items()}, str(root / "unet" / "diffusion_pytorch_model.safetensors"))
    save_file({k: (v * 0).contiguous() for k, v in usd.items()}, str(root / "unet" / "diffusion_pytorch_model.fp16.safetensors"))
    vcfg = gcfg.tiny_vae()
    (root / "vae" / "config.json").write_text(json.dumps({
        "_class_name": "AutoencoderKL", "block_out_channels": [32, 64, 64, 64], "sample_size": 64, "scaling_factor": 0.18215}))
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg))
    new_names = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}
    renamed = {}
    for k, v in vsd.items():
        for a, b in new_names.items():
            k = k.replace(a, b)
        renamed[k] = v.contiguous()
    save_file(renamed, str(root / "vae" / "diffusion_pytorch_model.safetensors"))

    unet = GyreHipUNet.from_pretrained(str(root / "unet"), torch_dtype=torch.bfloat16)
    assert unet.config == ucfg and unet.dtype == torch.bfloat16 and unet._source.endswith("unet")
    got = unet.state_dict()["conv_in.weight"].float()
    assert torch.allclose(got, usd["conv_in.weight"].to(torch.bfloat16).float())
    zero = GyreHipUNet.from_pretrained(str(root), subfolder="unet", variant="fp16")
    assert float(zero.state_dict()["conv_in.weight"].abs().max()) == 0
    vae = GyreHipVAE.from_pretrained(str(root / "vae"))
    assert vae.config.block_out_channels == (32, 64, 64, 64)
    assert torch.equal(vae.state_dict()["decoder.mid_block.attentions.0.query.weight"], vsd["decoder.mid_block.attentions.0.query.weight"])
    pipe = GyrePipeline.from_pretrained(str(root), device="cpu")
    assert isinstance(pipe.unet, GyreHipUNet) and isinstance(pipe.vae, GyreHipVAE) and pipe.text_encoder is None
    with pytest.raises(FileNotFoundError):
        GyrePipeline.from_pretrained(str(tmp_path / "nope"))
    # the engine's optional overrides (reference tests/engines.clip.yaml:12-17): 9-channel inpaint UNet + CLIP model folders
    import dataclasses
    from transformers import CLIPConfig, CLIPModel
    icfg = dataclasses.replace(ucfg, in_channels=9)
    (root / "inpaint_unet").mkdir()
    (root / "inpaint_unet" / "config.json").write_text(json.dumps({
        "_class_name": "UNet2DConditionModel", "in_channels": 9, "out_channels": 4, "sample_size": 16,
        "block_out_channels": [32, 64, 128, 128], "layers_per_block": 2, "cross_attention_dim": 64,
        "attention_head_dim": [2, 2, 4, 4], "norm_num_groups": 32,
        "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"]}))
    isd = weights.synthetic_state_dict(weights.unet_param_shapes(icfg))
    save_file({k: v.contiguous() for k, v in isd.items()}, str(root / "inpaint_unet" / "diffusion_pytorch_model.safetensors"))
    small = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2)
    clip = CLIPModel(CLIPConfig(text_config=dict(small, vocab_size=64, max_position_embeddings=16),
                                vision_config=dict(small, image_size=32, patch_size=8), projection_dim=16))
    clip.save_pretrained(str(root / "clip_model"))
    (root / "clip_model" / "preprocessor_config.json").write_text(json.dumps({
        "image_mean": [0.5, 0.5, 0.5], "image_std": [0.25, 0.25, 0.25], "size": {"shortest_edge": 32}}))
    full = GyrePipeline.from_pretrained(str(root), device="cpu", inpaint_unet=str(root / "inpaint_unet"),
                                        clip_model=str(root / "clip_model"))
    assert full.inpaint_unet.config.in_channels == 9 and full.clip_model is not None
    assert full.feature_extractor.image_mean == [0.5, 0.5, 0.5] and full.feature_extractor.size == {"shortest_edge": 32}
    px = torch.rand(2, 3, 32, 32)
    want = clip.eval().vision_model.embeddings(px)
    got = full.clip_model.vision_model.embeddings(px)           # patch embedding replaced by its matmul form: same values
    assert torch.allclose(got, want, atol=1e-5)
    # SDXL-style config keys
    xl = GyreHipUNet._config_from_json({
        "block_out_channels": [320, 640, 1280], "down_block_types": ["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
        "attention_head_dim": [5, 10, 20], "transformer_layers_per_block": [1, 2, 10], "cross_attention_dim": 2048,
        "use_linear_projection": True, "sample_size": 128, "addition_embed_type": "text_time",
        "addition_time_embed_dim": 256, "projection_class_embeddings_input_dim": 2816})
    assert xl == gcfg.sdxl_unet()
    with pytest.raises(NotImplementedError):
        GyreHipUNet._config_from_json({"class_embed_type": "timestep"})


def test_gemm4s_lds_swizzle_is_conflict_free():
    """kernels_gemm4s.hip: the LDS image written by the LDS-DMA pieces (lane-linear, k-slot XOR (row >> 1) & 7 on the
    source) is what the 32x32x16 fragment reads expect, and every ds_read_b128 lane group (MI355X_MICROARCH.md LDS
    table) touches 16 distinct 16-byte slots of the 256-byte bank row."""
    rows = 320
    image = {}                                      # LDS byte address -> (row, source k-slot)
    for piece in range(rows // 8):
        w = piece % 4
        for lane in range(64):
            kvs = (lane & 7) ^ ((4 * (w & 1) + (lane >> 4)) & 7)
            image[piece * 1024 + lane * 16] = (8 * piece + (lane >> 3), kvs)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for base in range(0, rows, 32):                 # a fragment = 32 consecutive rows starting at a multiple of 32
        for ks in range(4):
            addr = {}
            for lane in range(64):
                lo = (lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4)
                a = base * 128 + (lo ^ (ks << 5))
                addr[lane] = a
                assert image[a] == (base + (lane & 31), 2 * ks + (lane >> 5))      # row, logical k-slot
            for g in groups:
                slots = {(addr[l] % 256) // 16 for l in g}
                assert len(slots) == 16, (base, ks, g)


def test_hand_issued_loads_stay_untouched_until_their_wait():
    """Advisor (round 5): the small-problem GEMM requests its epilogue's bias / residual rows with hand-written
    `global_load_dwordx4` into "=v" outputs in FRONT of the LDS-DMA ring, so that the first counted wait covers them - but the
    compiler does not know those registers are pending: a copy, spill or re-materialisation it placed between the request and the
    `s_waitcnt vmcnt(0)` behind the K loop would read stale data silently.  The invariant is checked on the ISA of every build:
    between each such request and that wait no instruction reads or writes the destination registers (both storage flavours)."""
    import re
    import shutil
    import tempfile
    from gyre_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(B.CSRC, "kernels_gemm_sm.hip")
    checked = 0
    for extra in ([], ["-DGYRE_STORE_F16"]):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            r = subprocess.run([hipcc, *B.FLAGS, *extra, "--offload-device-only", "-S", src, "-o", out], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            text = open(out).read()
        # one block of lines per kernel; the scan follows the control flow (a register initialisation on the path that does NOT issue
        # the requests sits textually between a request and the wait without being reachable from it)
        for m in re.finditer(r"^(_Z9k_gemm_sm\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
            lines = [ln.split(";")[0].strip() for ln in m.group(2).splitlines()]
            lines = [ln for ln in lines if ln and not (ln.startswith(".") and not ln.endswith(":"))]
            label_at = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
            first_barrier = next(i for i, ln in enumerate(lines) if ln.startswith("s_barrier"))
            wait = next(i for i, ln in enumerate(lines) if i > first_barrier and re.fullmatch(r"s_waitcnt vmcnt\(0\)", ln))
            loads = [(i, ln) for i, ln in enumerate(lines[:first_barrier]) if ln.startswith("global_load_dwordx4") and "lds" not in ln]
            assert loads, "the hand-issued epilogue requests are gone: update this test with the kernel"

            def reachable(start):
                seen, todo = set(), [start]
                while todo:
                    i = todo.pop()
                    while i < len(lines) and i not in seen and i != wait:
                        seen.add(i)
                        ln = lines[i]
                        br = re.match(r"s_(c?branch\w*)\s+(\S+)", ln)
                        if br:
                            todo.append(label_at[br.group(2)])
                            if br.group(1) == "branch":
                                break
                        i += 1
                return seen

            for i, ln in loads:
                a, b = map(int, re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", ln).groups())
                dest = set(range(a, b + 1))
                path = reachable(i + 1)
                assert path, "no path from the request to the wait?"
                for j in sorted(path):
                    used = set(int(x) for x in re.findall(r"\bv(\d+)\b", lines[j]))
                    for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", lines[j]):
                        used |= set(range(int(lo), int(hi) + 1))
                    assert not (used & dest), f"{m.group(1)}: `{lines[j]}` touches v[{a}:{b}] between its request and the wait"
                checked += 1
    assert checked >= 8          # two kernels x (2 + 4) requests x two flavours, at least
