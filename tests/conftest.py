import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """GYRE_TEST_WORKERS=N runs the suite on N pytest-xdist workers, one test FILE per worker at a time (module-level model caches
    stay valid).  OPT-IN only: measured on a 256-thread GPU box, four workers were SLOWER than the serial run (75 % of the -m gpu
    suite in 25 min against 23 min for all of it) - the wall time is the fp32 CPU oracle, whose ATen threads the workers then
    fight over; the cure for the suite's length is cheaper oracle work, not more processes."""
    want = os.environ.get("GYRE_TEST_WORKERS")
    if not want or want == "0" or os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):
        return None                                          # (a worker re-enters this hook with the controller's options: never nest)
    try:
        import xdist  # noqa: F401
    except Exception:
        return None
    opt = config.option
    if getattr(opt, "numprocesses", None) or getattr(opt, "collectonly", False):
        return None
    opt.numprocesses = int(want)
    opt.dist = "loadfile"
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fp32 CPU oracle is most of the suite's wall time, and ATen's default of one thread per core makes it SLOWER on the GPU
    # boxes (256 logical CPUs, shared): the same three full-size parity tests took 13.6 s with 16 threads, 15.0 with 32, 19.9 with
    # 64 and 37.3 with 128 (7.7 CPU-minutes instead of 1.1) - profiles/README.md, round 5.  GYRE_TEST_THREADS overrides.
    import torch
    torch.set_num_threads(max(1, min(int(os.environ.get("GYRE_TEST_THREADS", "16")), os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")))


@pytest.fixture(autouse=True)
def _gpu_keepalive(request):
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gpu_util
        gpu_util.release_kept()
