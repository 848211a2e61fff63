import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

WSEED, ISEED, NSEED = 11, 7, 1234   # seeds used when the golden vectors were recorded


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load


def rel_err(a, b):
    """max |a-b| / max |b| -- the per-tensor relative tolerance used by every parity test."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
