#!/usr/bin/env python3
"""Random starts of StandardLinearModel.fit at config 2's shape (N=1M, D=32, F=4096): ranked by the objective-only
evaluation (statistics pass + posterior) vs by full `_elbo` evaluations as the reference does."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd import StandardLinearModel
rng = np.random.default_rng(0)
N, d = 1_000_000, 32
X = rng.standard_normal((N, d), dtype=np.float32)
y = (np.sin(X @ rng.standard_normal(d).astype(np.float32)) + 0.1 * rng.standard_normal(N, dtype=np.float32)).astype(np.float32)
NSTARTS = 30
for mode in ("objective-only", "full _elbo"):
    b = bs.RandomRBF(nbases=2048, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    slm = StandardLinearModel(b, nstarts=NSTARTS, maxiter=0, random_state=3)
    if mode == "full _elbo":
        StandardLinearModel._elbo_objective = lambda self, Xa, ya, *a: self._elbo(Xa, ya, *a)[0]
    t0 = time.perf_counter()
    slm.fit(X, y)
    dt = time.perf_counter() - t0
    print("%-15s %d random starts + maxiter=0: %.2f s" % (mode, NSTARTS, dt), flush=True)

# --- GLM (config 5's shape): 40 random starts, batch 65 536, device sampler
from revrand_amd import GeneralizedLinearModel
import revrand_amd.likelihoods as lk
import revrand_amd.optimize as opt
X = X[:, :32]
yc = np.random.RandomState(3).poisson(np.exp(0.3 * X[:, 0])).astype(float)
for mode in ("objective-only", "full step"):
    b = bs.RandomRBF(nbases=1024, Xdim=32, random_state=1, lenscale=Parameter(np.ones(32), Positive()))
    glm = GeneralizedLinearModel(lk.Poisson(), b, K=10, nsamples=50, random_state=2, maxiter=1, batch_size=65536,
                                 sampler="device", nstarts=40)
    if mode == "full step":
        orig = GeneralizedLinearModel._elbo
        GeneralizedLinearModel._elbo = lambda self, *a, objective_only=False: (orig(self, *a)[0] if objective_only else orig(self, *a))
    t0 = time.perf_counter()
    glm.fit(X, yc)
    print("GLM %-15s 40 random starts + 1 step: %.2f s" % (mode, time.perf_counter() - t0), flush=True)
