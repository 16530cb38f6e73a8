"""Where the host's time goes in a default-shaped GLM fit through the fused loop with the reference's random stream."""
import cProfile
import os
import pstats
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
r2 = np.random.RandomState(1)
out = np.empty(41500, dtype=np.float32)
_hip.legacy_randn(r2, 41500, np.float32, out=out)
t0 = time.perf_counter()
for _ in range(200):
    _hip.legacy_randn(r2, 41500, np.float32, out=out)
print("legacy_randn(41500): %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))


def fit():
    basis = bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=20, Xdim=2) + bs.RandomMatern52(nbases=20, Xdim=2)
    glm = GeneralizedLinearModel(lk.Gaussian(), basis, random_state=1, sampler=os.environ.get("SAMPLER", "host"))
    np.random.seed(0)
    glm.fit(X, y)


fit()
pr = cProfile.Profile()
pr.enable()
fit()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
