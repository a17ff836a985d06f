"""The oracle's UNet ASSEMBLY against the reference's vendored block forwards (build container only).

nonfree/tome_unet.py:34-221 re-states, inside the reference tree, the forward bodies of diffusers' CrossAttnDownBlock2D,
UNetMidBlock2DCrossAttn, CrossAttnUpBlock2D, SpatialTransformer, BasicTransformerBlock and CrossAttention (their ToMe subclasses
with r = 0), and gyre/pipeline/models/memory_efficient_cross_attention.py:32-60 the head split.  tests/golden/
ref_block_wiring_probe.py EXECUTES those bodies with the oracle's leaf functions as their sub-modules and rebuilds the whole
trunk; every level of the result must equal oracle/models_ref.py's own forward (fp32 rounding only).  This pins the wiring -
block order, skip tuple bookkeeping, concat order, transformer residual / reshape / sub-layer order, attention arithmetic - to
code the reference ships; the leaves' internals (ResnetBlock2D, samplers, GEGLU, timestep embedding) remain unpinned."""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")


def test_oracle_assembly_equals_the_reference_block_forwards():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "ref_block_wiring_probe.py")], env=env,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if "PROBE_JSON " in l]
    assert r.returncode == 0 and line, (r.stdout + r.stderr)[-3000:]
    out = json.loads(line[0].split("PROBE_JSON ", 1)[1])
    tol = 5e-5                                             # activations are O(1-10): fp32 reassociation only
    assert out["cross_attention_max_abs"] <= tol and out["transformer_block_max_abs"] <= tol
    assert out["spatial_transformer_max_abs"] <= tol
    assert set(out["levels_max_abs"]) == {"down0", "down1", "down2", "down3", "mid", "up0", "up1", "up2", "up3"}
    assert max(out["levels_max_abs"].values()) <= tol
    assert out["unet_out_max_abs"] <= tol and out["unet_out_absmax"] > 0.5 and out["skips_consumed"]
    # ControlNet residual injection: the reference's own patcher (controlnet/unet_patcher.py:60-95 UNet2DConditionModelHook.pre_forward,
    # UpBlockWrapper, MidBlockWrapper) rewires that trunk; the oracle's down_res / mid_res must give the same output
    assert out["controlnet_patcher_max_abs"] <= tol and out["controlnet_effect"] > 0.1
    # T2I-adapter states: t2i_adapter/unet_patcher.py:21-86 (DownsamplerWrapper / CrossAttnDownBlock2DHook / DownBlockWrapper, with
    # their in-place `+=` that also changes the skip connection just stored) vs the oracle's adapter_states
    assert out["t2i_patcher_max_abs"] <= tol and out["t2i_effect"] > 0.1
