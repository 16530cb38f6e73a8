#!/usr/bin/env python3
"""Config 1 of BASELINE.json on the GPU path: StandardLinearModel.fit, RandomRBF nbases=256, D=8,
N=10k, nstarts=0, maxiter=20 (the reference's NumPy path took 21.1 s / 56 `_elbo` calls on the
survey box).  Also times one resident `_elbo` at a larger shape."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd.basis_functions import RandomRBF
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel

rs = np.random.RandomState(0)
N, d, n = 10000, 8, 256
X = rs.randn(N, d)
y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
calls = [0]
orig = StandardLinearModel._elbo
def counted(self, *a):
    calls[0] += 1
    return orig(self, *a)
StandardLinearModel._elbo = counted
for rep in range(2):
    calls[0] = 0
    slm = StandardLinearModel(RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(1.0, Positive())),
                              var=Parameter(1.0, Positive()), nstarts=0, maxiter=20)
    t0 = time.perf_counter(); slm.fit(X, y); dt = time.perf_counter() - t0
    Ey = slm.predict(X[:1000])
    print("C1 fit rep%d: %.3f s, %d _elbo calls (%.1f ms each), smse %.4f" % (
        rep, dt, calls[0], 1e3 * dt / calls[0], ((Ey - y[:1000]) ** 2).mean() / y.var()))

# one resident _elbo at N=200k, d=32, F=2048
N, d, n = 200000, 32, 1024
X = rs.randn(N, d).astype(np.float32); y = np.sin(X @ rs.randn(d)).astype(np.float32)
b = RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
slm = StandardLinearModel(b); slm.obj_ = -np.inf
slm._state = b.device_fit_state(X, y)
for rep in range(2):
    t0 = time.perf_counter(); G, bv, yty = slm._state.gram(np.ones(d)); t1 = time.perf_counter()
    o = slm._elbo(X, y, 0.5, 1.0, np.ones(d)); t2 = time.perf_counter()
    print("resident _elbo N=%d F=%d ARD d=%d: stats pass %.3f s, whole _elbo %.3f s" % (N, 2 * n, d, t1 - t0, t2 - t1))
slm._state.release()
