"""The reference's default GLM usage (tests/test_models.py:83-147: batch_size=10, K=10, nsamples=50, maxiter=3000, nstarts=500, a
concatenation) through the fused small-batch loop (rr_glm_svi): seconds per fit, and where they go -- the batched random
starts, the launches (device time per step, measured by synchronising behind each), the host's part.
SAMPLER=device|host, FUSED=1|0, SYNC=1 (time each launch), REPS."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs  # noqa: E402
from revrand_amd import _hip  # noqa: E402
from revrand_amd import likelihoods as lk  # noqa: E402
from revrand_amd.glm import GeneralizedLinearModel  # noqa: E402

rs = np.random.RandomState(100)
x = np.linspace(-5, 5, 600)
y = 3 + 2 * x + rs.randn(600) * 1e-4
X = np.column_stack((np.ones(600), x))
sampler = os.environ.get("SAMPLER", "device")
fused = os.environ.get("FUSED", "1") == "1"
sync = os.environ.get("SYNC", "0") == "1"
real_run, real_starts = _hip.FusedSvi.run, _hip.FusedSvi.starts
marks = {"run": [], "starts": []}


def run(self, n, *a, **k):
    t0 = time.perf_counter()
    r = real_run(self, n, *a, **k)
    if sync:
        self.dev.sync()
    marks["run"].append((n, time.perf_counter() - t0))
    return r


def starts(self, *a, **k):
    t0 = time.perf_counter()
    r = real_starts(self, *a, **k)
    marks["starts"].append(time.perf_counter() - t0)
    return r


_hip.FusedSvi.run, _hip.FusedSvi.starts = run, starts
for rep in range(int(os.environ.get("REPS", "3"))):
    marks["run"].clear(); marks["starts"].clear()
    basis = bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=20, Xdim=2) + bs.RandomMatern52(nbases=20, Xdim=2)
    glm = GeneralizedLinearModel(lk.Gaussian(), basis, random_state=1, sampler=sampler)
    glm._fused_sgd = fused
    np.random.seed(0)
    t0 = time.perf_counter()
    glm.fit(X, y)
    t = time.perf_counter() - t0
    Ey = glm.predict(X[:50])
    print("%s sampler, %s: fit %.3f s, smse %.2e" % (sampler, "fused" if fused else "resident", t, ((Ey - y[:50]) ** 2).mean() / y.var()))
    if marks["starts"]:
        print("   random starts call(s): %s ms" % ["%.1f" % (1e3 * v) for v in marks["starts"]])
    if marks["run"]:
        n = sum(m[0] for m in marks["run"])
        tt = sum(m[1] for m in marks["run"])
        print("   %d launches, %d steps, %.1f ms inside run()%s: %.1f us per step"
              % (len(marks["run"]), n, 1e3 * tt, " + sync" if sync else "", 1e6 * tt / n))
