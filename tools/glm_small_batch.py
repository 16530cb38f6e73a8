"""The reference's default GLM usage -- batch_size=10, maxiter=3000, a concatenation (tests/test_models.py:83-147) -- through the
resident SVI loop and through the host loop: seconds per fit, steps per second."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs  # noqa: E402
from revrand_amd import likelihoods as lk  # noqa: E402
from revrand_amd.glm import GeneralizedLinearModel  # noqa: E402

rs = np.random.RandomState(100)
x = np.linspace(-5, 5, 600)
y = 3 + 2 * x + rs.randn(600) * 1e-4
X = np.column_stack((np.ones(600), x))
for resident in (True, False, True, False):
    basis = bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=20, Xdim=2) + bs.RandomMatern52(nbases=20, Xdim=2)
    glm = GeneralizedLinearModel(lk.Gaussian(), basis, random_state=1, sampler=os.environ.get("SAMPLER", "host"))
    glm._resident_sgd = resident
    marks = []
    ah = glm._ahead

    def ahead(batch, draws=True, ah=ah):
        t0 = time.perf_counter(); r = ah(batch, draws); marks.append(time.perf_counter() - t0); return r
    glm._ahead = ahead
    from revrand_amd import _hip
    calls = []
    if not hasattr(_hip.ResidentSgd, "_real_step"):
        _hip.ResidentSgd._real_step = _hip.ResidentSgd.step

    def timed(self, *a, calls=calls, **k):
        t0 = time.perf_counter(); r = _hip.ResidentSgd._real_step(self, *a, **k); calls.append(time.perf_counter() - t0); return r
    _hip.ResidentSgd.step = timed
    np.random.seed(0)
    t0 = time.perf_counter()
    glm.fit(X, y)
    t = time.perf_counter() - t0
    Ey = glm.predict(X[:50])
    print("%s: fit %.2f s (%d random starts + %d steps: %.0f us per step), smse %.2e"
          % ("resident loop" if resident else "host loop    ", t, glm.nstarts, glm.maxiter, 1e6 * t / (glm.maxiter + glm.nstarts),
             ((Ey - y[:50]) ** 2).mean() / y.var()))
    if calls:
        print("   rr_glm_sgd_step on the host: median %.0f us" % (1e6 * np.median(calls)))
    if marks:
        print("   upload stage per batch: median %.0f us" % (1e6 * np.median(marks)))
    ck = glm.__dict__.get("_resident_clock")
    if ck is not None and len(ck) > 24:   # (the step-per-call loop's per-step clock; the fused loop records one time per launch)
        dt = 1e6 * np.diff(ck[20:-2])
        print("   step intervals: mean %.0f us, median %.0f us, p90 %.0f us" % (dt.mean(), np.median(dt), np.percentile(dt, 90)))
