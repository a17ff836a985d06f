"""gyre_amd.clipguided against the REFERENCE's own ClipGuidedMode.

tests/golden/ref_clipguided_probe.py imports gyre/pipeline/unet/clipguided.py from /root/reference (absent torchvision /
k_diffusion / ResizeRight pieces replaced by functional stand-ins) and drives both implementations over six steps of a toy
differentiable UNet / VAE / tiny CLIP with the same per-image generators: denoised predictions (k-diffusion path,
clipguided.py:267-299) and noise predictions (diffusers-style path, :180-216), loss history, flat-loss stop and the
generators' final state must agree.  Build container only."""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")


def test_clip_guided_mode_matches_the_reference_class():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "ref_clipguided_probe.py")], env=env,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if "PROBE_JSON " in l]
    assert r.returncode == 0 and line, (r.stdout + r.stderr)[-3000:]
    out = json.loads(line[0].split("PROBE_JSON ", 1)[1])
    cases = {k: v for k, v in out.items() if isinstance(v, dict)}
    assert len(cases) >= 9
    for name, c in cases.items():
        tol = 0.0 if name.startswith("k_") else 1e-6 * max(c["ref_absmax"], 1.0)     # d-path: python-float vs tensor scalars
        assert c["max_abs_diff"] <= tol, (name, c)
        assert c["lossavg_diff"] <= 1e-6 and c["n_loss"][0] == c["n_loss"][1], (name, c)
        assert c["flat"][0] == c["flat"][1] and c["gen_state_equal"], (name, c)
        assert c["guidance_effect"] > 1e-3, (name, c)                       # the guided result differs from plain CFG
    assert cases["k_flatloss_stops"]["flat"] == [True, True] and cases["k_flatloss_stops"]["n_loss"] == [3, 3]
    assert out["noc_with_cutouts_ref"] == out["noc_with_cutouts_mine"] == "RuntimeError"
