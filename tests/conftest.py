"""Shared pytest plumbing: the ``gpu`` marker, golden-fixture loader, oracle import."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def normwise(a, b):
    """max|a-b| / max|b| -- Phi entries cross zero, so tolerances are normwise."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    den = np.abs(b).max() if b.size else 1.0
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0)) if b.size else 0.0


@pytest.fixture(autouse=True)
def _seed_global_numpy_rng():
    """Estimators built with random_state=None draw their start points from NumPy's global RandomState
    (check_random_state(None)); seed it per test so that a run does not depend on test order or on chance."""
    np.random.seed(20260928)
    yield
