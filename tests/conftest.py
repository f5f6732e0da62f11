import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU parity oracle (test infrastructure)."""
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def chamfer_golden():
    blob = np.load(os.path.join(GOLDEN, "chamfer_golden.npz"))
    cases = {}
    for key in blob.files:
        name, field = key.split("/")
        cases.setdefault(name, {})[field] = blob[key]
    return cases


def rand_clouds(seed, *shape):
    return np.random.default_rng(seed).random(shape, dtype=np.float32)
