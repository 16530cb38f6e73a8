#!/usr/bin/env python3
"""Kernel mix of one resident `_elbo` at a mid-size width (where the N x F x F GEMMs no longer hide everything else):
    rocprofv3 --kernel-trace --stats -- python tools/elbo_mix.py [N] [nbases] [d]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel as SLM
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rng = np.random.default_rng(0)
X = rng.standard_normal((N, d), dtype=np.float32)
y = np.sin(X[:, 0]).astype(np.float32)
b = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
s = SLM(b); s.obj_ = -np.inf; s._state = s._make_state(X, y)
ls = np.linspace(0.8, 1.5, d)
s._elbo(X, y, 0.3, 1.0, ls)
t0 = time.perf_counter()
for _ in range(5):
    s._elbo(X, y, 0.3, 1.0, ls)
print("one _elbo at N=%d F=%d d=%d: %.2f ms" % (N, 2 * n, d, (time.perf_counter() - t0) / 5 * 1e3))
s._state.release()
