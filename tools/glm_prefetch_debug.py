"""Run-to-run and prefetch-vs-not comparison of short GLM fits (debug aid for tests/test_gpu_glm.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd import likelihoods as lk
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.glm import GeneralizedLinearModel as GLM
rs = np.random.RandomState(5)
N, d = 6000, 5
X = rs.randn(N, d)
f = 0.8 * np.sin(X[:, 0]) + 0.3 * X[:, 1]
cases = [("poisson", lk.Poisson(), rs.poisson(np.exp(f)).astype(float), ()),
         ("binomial", lk.Binomial(), rs.binomial(7, 1 / (1 + np.exp(-f))).astype(float), (7 * np.ones(N),)),
         ("gaussian", lk.Gaussian(), f + 0.1 * rs.randn(N), ())]
def nw(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
for name, lik, y, largs in cases:
    fits = {}
    for tag, pf in (("off1", "0"), ("off2", "0"), ("on1", "1"), ("on2", "1")):
        os.environ["RR_GLM_BATCH_PREFETCH"] = pf
        basis = bs.RandomRBF(nbases=40, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) + bs.LinearBasis(onescol=True)
        glm = GLM(lik, basis, K=2, nsamples=8, batch_size=500, maxiter=int(os.environ.get("ITERS", 12)), nstarts=2, random_state=3)
        glm.fit(X, y, likelihood_args=largs)
        fits[tag] = np.concatenate((glm.weights_.ravel(), glm.covariance_.ravel()))
    print(name, "off1-off2 %.2e  on1-on2 %.2e  off1-on1 %.2e" % (nw(fits["off1"], fits["off2"]), nw(fits["on1"], fits["on2"]), nw(fits["off1"], fits["on1"])), flush=True)
