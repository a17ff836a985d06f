"""-m gpu: randomised race / determinism screen of the counted-vmcnt LDS-DMA kernels (tools/race_screen.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [0, 7])
def test_race_screen(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_screen.py"), str(seed)], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "problems = 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
