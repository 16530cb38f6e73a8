"""Config 1's `_elbo` (RandomRBF nbases=256, D=8, N=10k, resident) K times in a row: what rocprofv3 --kernel-trace / --hip-trace
counts per evaluation (launches, synchronising calls), and its wall-clock.  `python tools/c1_latency.py [f32|f64] [K]`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
r = np.random.RandomState(11)
X = r.randn(10000, 8)
y = np.sin(X @ np.array([1.0, -0.7, 0.5, 0.3, -0.2, 0.9, -0.4, 0.1])) + 0.1 * r.randn(10000)
b = bs.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(2.0, Positive()), regularizer=Parameter(10.0, Positive()), dtype=dtype)
slm = StandardLinearModel(b)
slm.obj_ = -np.inf
slm._defer_cov = True
slm._state = slm._make_state(X, y)
slm._elbo(X, y, 0.02, 10.0, 2.0)
t0 = time.perf_counter()
for k in range(K):
    slm._elbo(X, y, 0.02, 10.0, 2.0 * (1 + 1e-6 * k))
dt = (time.perf_counter() - t0) / K
print("%s: %.3f ms per _elbo over %d evaluations" % (dtype, 1e3 * dt, K))
slm._state.release()
