#!/usr/bin/env python3
"""Second _elbo pass at the headline width (F=4096, D=32): kernel times come from rocprofv3 --kernel-trace."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd.basis_functions import RandomRBF
from revrand_amd.btypes import Parameter, Positive
N, d, n = 500_000, 32, 2048
rng = np.random.default_rng(0)
X = rng.standard_normal((N, d), dtype=np.float32)
y = np.sin(X @ rng.standard_normal(d, dtype=np.float32)).astype(np.float32)
b = RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
st = b.device_fit_state(X, y)
F = 2 * n
m = rng.standard_normal(F) * 0.01
A = rng.standard_normal((F, 64)); C = A @ A.T / 64 + np.eye(F)
for rep in range(2):
    t0 = time.perf_counter(); sq, dh = st.second_pass(np.ones(d), m, C, 0.5); dt = time.perf_counter() - t0
    print("second pass N=%d F=%d: %.3f s wall (incl. C upload), sqErr %.4g, |dhyp| %.3g" % (N, F, dt, sq, np.abs(dh).max()))
print("gemm flops", 2.0 * N * F * F)
