"""the native module classes are selected and loaded by the REFERENCE's own engine manager code - executed,
not just asserted in INTEGRATION.md.  tests/golden/ref_manager_probe.py imports gyre/manager.py from /root/reference (absent
third-party roots stubbed as the reference stubs them itself) in a subprocess and runs, unmodified:
  EngineManager._import_class          manager.py:1024-1066   `class: gyre_amd.modules.GyreHipUNet`
  EngineManager._parse_class_details   manager.py:1114-1143   `pkg.Class/factory(arg=v)`
  EngineManager._load_model_from_weights  manager.py:1145-1252   from_pretrained(weight_path[/unet], torch_dtype=float16?, variant="fp16"?)
  EngineManager._load_module_fallback  manager.py:1068-1112   Class(**config) + load_state_dict + eval
  model_utils.clone_model              pipeline/model_utils.py:172-259   per-device-slot clone sharing the CPU master weights
Build container only (the reference tree does not travel to the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")


def test_reference_manager_loads_and_clones_native_modules():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "ref_manager_probe.py")], env=env,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("PROBE_JSON ")]
    assert r.returncode == 0 and line, (r.stdout + r.stderr)[-3000:]
    out = json.loads(line[0][len("PROBE_JSON "):])
    assert out["import_class"] and out["unet_type"] == "GyreHipUNet" and out["unet_source"]
    assert out["unet_keys_equal"] and out["unet_values_equal"] and out["unet_eval"]
    assert out["unet_config"] == [4, 16, [32, 64, 128, 128]]                  # config attributes the pipeline reads
    assert out["vae_fp16_dtype"] == "torch.float16" and out["vae_block_out_channels"] == [32, 64, 64, 64]
    assert out["fallback_values_equal"]
    assert out["class_details"] == ["gyre_amd.modules.GyreHipUNet", "from_pretrained", {"variant": "fp16"}]
    assert out["clone_type"] == "GyreHipUNet" and out["clone_keys_equal"] and out["clone_shares_storage"] and out["clone_has_config"]
    assert out["modules_walk"]
