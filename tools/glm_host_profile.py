#!/usr/bin/env python3
"""Where does the host time of a config-5 SVI step go?  cProfile over fit() with the device sampler."""
import cProfile, pstats, os, sys, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd import likelihoods as lk
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.glm import GeneralizedLinearModel
logging.getLogger("revrand_amd").setLevel(logging.ERROR)
N, d, n, K, L, M = 1_000_000, 32, 1024, 10, 50, 65536
rng = np.random.default_rng(5)
X = rng.standard_normal((N, d), dtype=np.float32)
y = rng.poisson(np.exp(0.3 * X[:, 0].astype(np.float64))).astype(np.float64)
sampler = sys.argv[1] if len(sys.argv) > 1 else "device"
def run(iters):
    g = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                               K=K, nsamples=L, batch_size=M, maxiter=iters, nstarts=0, random_state=2, sampler=sampler)
    g.fit(X, y)
run(4)
pr = cProfile.Profile()
pr.enable()
run(60)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
