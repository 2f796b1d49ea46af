import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    return np.load(os.path.join(GOLD, "hotpath.npz"))


@pytest.fixture(scope="session")
def fields():
    from oracle import oracle as orc
    fa = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_A.npz"))
    fb = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_B.npz"), shared=fa)
    # shared nets only: planes of B are its own
    return {"A": fa, "B": fb}


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
