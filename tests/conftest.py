import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
