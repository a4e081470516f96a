import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture
def golden():
    return load_golden


def hashed_uniform(shape, salt, lo=-1.0, hi=1.0):
    """Same exact-integer-hash generator as tests/golden/gen_golden.py (inputs too big to store)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64) + np.uint64(salt) * np.uint64(0x9E3779B1)
    i = (i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    i ^= i >> np.uint64(15)
    i = (i * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    i ^= i >> np.uint64(13)
    u = (i >> np.uint64(8)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)
