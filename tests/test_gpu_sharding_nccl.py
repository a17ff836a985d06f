"""The RCCL leg of the data-parallel path on the ONE GPU the test box has (-m gpu): two torch.distributed ranks share
cuda:0, split a 3-image request 2 + 1 (reference split rule, services/generate.py:977-990), run the whole loop on the
native UNet and all_gather the finished latents over the "nccl" backend.  With batch-invariant planning the gathered
result must be bit-identical to the single-process run.  Some RCCL builds refuse two ranks on one device - then the
test is skipped with RCCL's own message (the gloo twin, tests/test_sharding_gloo.py, always runs)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generate_sharded_over_rccl_two_ranks_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tests", "_nccl_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr
    if "RCCL_REFUSED:" in out:
        pytest.skip("RCCL refuses two ranks on one GPU: " + out.split("RCCL_REFUSED:", 1)[1].splitlines()[0])
    assert r.returncode == 0, out[-3000:]
    assert "RESULT_OK" in out, out[-3000:]
