import os, sys, time, logging, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd import likelihoods as lk
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.glm import GeneralizedLinearModel
logging.getLogger("revrand_amd").setLevel(logging.ERROR)
N, d, n, K, L, M = 2_000_000, 32, 1024, 10, 50, 65536
rng = np.random.default_rng(5)
X = rng.standard_normal((N, d), dtype=np.float32)
y = rng.poisson(np.exp(0.3 * X[:, 0].astype(np.float64))).astype(np.float64)
def run(iters, prof=None):
    g = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                               K=K, nsamples=L, batch_size=M, maxiter=iters, nstarts=0, random_state=2)
    t = time.perf_counter()
    if prof: prof.enable()
    g.fit(X, y)
    if prof: prof.disable()
    return time.perf_counter() - t
run(4)
t8 = run(8); t48 = run(48)
print("fit step %.2f ms" % (1e3 * (t48 - t8) / 40))
p = cProfile.Profile()
run(108, p)
st = pstats.Stats(p); st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
