import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The product's default arithmetic is bf16x3 (modelzoo/_factory.PRODUCT_DEFAULT_PRECISION).  The tests that build their models through
# the zoo factories / the drivers without naming an arithmetic are the exact-product parity tests -- tolerances written for fp32
# MFMA products against the oracle -- so the test session asks for 'f32' explicitly, here; the tests of the other arithmetics name
# theirs, and tests/test_gpu_runner.py::test_the_drivers_default_arithmetic_* removes this variable to see the product default.
os.environ.setdefault("ADN_PRECISION", "f32")


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
