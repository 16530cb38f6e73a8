"""`_elbo` evaluations of StandardLinearModel at config 2's width (RandomRBF F=4096, D=32, N rows, ARD), device
posterior vs host posterior (RR_POSDEF=host): wall time per evaluation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip
from revrand_amd.basis_functions import RandomRBF
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel
N, d, n = int(os.environ.get("ROWS", 1_000_000)), 32, int(os.environ.get("NBASES", 2048))
rng = np.random.default_rng(0)
X = rng.standard_normal((N, d), dtype=np.float32)
y = np.sin(X @ rng.standard_normal(d, dtype=np.float32)).astype(np.float32) + 0.1 * rng.standard_normal(N, dtype=np.float32)
DT = os.environ.get("DTYPE", "f32")
b = RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()), dtype=DT)
slm = StandardLinearModel(b)
slm.obj_ = -np.inf
slm._state = b.device_fit_state(X, y)
print("posterior on device:", _hip.posterior_available(), "arithmetic:", DT, flush=True)
ls = np.ones(d)
for rep in range(3):
    t0 = time.perf_counter()
    f, g = slm._elbo(X, y, 0.5, 1.0, ls * (1 + 0.01 * rep))
    print("N=%d F=%d _elbo %.3f s  (-ELBO %.6g)" % (N, 2 * n, time.perf_counter() - t0, f), flush=True)
slm._state.release()
