#!/usr/bin/env python3
"""Wall-clock per SVI step of fit() at config 5's shape (N = 1M resident): (time of 260 steps - time of 60 steps) / 200,
for the reference's random stream ("host") and the device sampler.  SWITCH=<seconds> sets the interpreter's
thread switch interval (no measurable effect: the run-to-run spread of the host-sampler number, 5.4 - 10 ms on one box,
is larger than anything these switches move)."""
import os, sys, time, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd import likelihoods as lk
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.glm import GeneralizedLinearModel
logging.getLogger("revrand_amd").setLevel(logging.ERROR)
if os.environ.get("SWITCH"):
    sys.setswitchinterval(float(os.environ["SWITCH"]))
N, d, n, K, L, M = 1_000_000, 32, 1024, 10, 50, 65536
rng = np.random.default_rng(5)
X = rng.standard_normal((N, d), dtype=np.float32)
y = rng.poisson(np.exp(0.3 * X[:, 0].astype(np.float64))).astype(np.float64)
def run(sampler, iters):
    g = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                               K=K, nsamples=L, batch_size=M, maxiter=iters, nstarts=0, random_state=2, sampler=sampler)
    t = time.perf_counter(); g.fit(X, y); return time.perf_counter() - t
for sampler in sys.argv[1:] or ["host", "device"]:
    run(sampler, 5)
    res = []
    for rep in range(3):
        t60, t260 = run(sampler, 60), run(sampler, 260)
        res.append((t260 - t60) / 200 * 1e3)
    print("%s sampler: %s ms per step" % (sampler, ", ".join("%.2f" % r for r in res)))
