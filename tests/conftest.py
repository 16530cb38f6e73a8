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


def pytest_collection_modifyitems(config, items):
    """Every GPU test is bounded: a device call that never returns (or a rank of a multi-process test that never exits)
    must end the run with every thread's stack on stderr, not hold the GPU box until somebody else's limit.  The "thread"
    method, because a main thread parked inside a ctypes call never gets to run a signal handler."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("gpu") is None:
            continue
        m = it.get_closest_marker("timeout")
        if m is None:
            it.add_marker(pytest.mark.timeout(1500, method="thread"))
        elif "method" not in m.kwargs and len(m.args) < 2:  # a test's own (longer) limit, the same method
            it.add_marker(pytest.mark.timeout(m.args[0] if m.args else m.kwargs.get("timeout", 1500), method="thread"), append=False)


def pytest_sessionfinish(session, exitstatus):
    """Nothing this run started may outlive it: a worker process that delivered its result and then never exits (multi-process
    tests share the one GPU of the box) would be joined by multiprocessing's exit handler -- forever -- and keeps the run's
    stdout open meanwhile.  Reusable joblib workers (GridSearchCV(n_jobs=2)) hold a device context each: ended here too."""
    import multiprocessing
    try:
        from joblib.externals.loky import get_reusable_executor
        get_reusable_executor().shutdown(wait=False, kill_workers=True)
    except Exception:
        pass
    for ctxname in ("fork", "spawn", "forkserver"):
        try:
            kids = multiprocessing.get_context(ctxname).active_children()
        except Exception:
            continue
        for p in kids:
            sys.stderr.write("conftest: ending a child process left running by the tests: %r\n" % (p,))
            p.kill()
            p.join(10)


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
