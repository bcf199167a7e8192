import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _seed_global_numpy_rng(request):
    """The model-zoo factories draw their initial weights from NumPy's global generator, as Lasagne's initialisers do (the
    reference never seeds it).  A test must not depend on the draw: every test starts from a seed derived from its own id."""
    import zlib
    import numpy as np
    np.random.seed(zlib.crc32(request.node.nodeid.encode()) & 0xFFFFFFFF)
    yield
