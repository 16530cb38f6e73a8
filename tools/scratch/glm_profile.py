import cProfile, pstats, sys, os, io
sys.path.insert(0, ".")
import numpy as np
import revrand_amd.basis_functions as bs
from revrand_amd import likelihoods as lk
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.glm import GeneralizedLinearModel
rng = np.random.default_rng(0)
N, d = 500_000, 32
X = rng.standard_normal((N, d))
y = rng.poisson(np.exp(0.6 * np.sin(X[:, 0]))).astype(np.float64)
basis = bs.RandomRBF(nbases=1024, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
glm = GeneralizedLinearModel(lk.Poisson(), basis, K=10, nsamples=50, batch_size=65536, maxiter=20, nstarts=2, random_state=2)
glm.fit(X, y)
pr = cProfile.Profile(); pr.enable()
glm.fit(X, y)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
