#!/usr/bin/env python3
"""predict_moments / predict of config 3's concatenation (RandomMatern52 n = 4096 + LinearBasis, F_tot = 8257, D = 64) for a
300 000-row host query: ms per call and the fraction of the f32 MFMA peak on F^2 flop per row (triangular factor)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel
N, d, n = int(os.environ.get("ROWS", 300000)), 64, 4096
rng = np.random.default_rng(7)
X = rng.standard_normal((N, d), dtype=np.float32)
y = np.sin(X[:, 0]) + 0.1 * rng.standard_normal(N, dtype=np.float32)
basis = bs.RandomMatern52(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.full(d, 3.0), Positive())) + bs.LinearBasis(onescol=True)
slm = StandardLinearModel(basis, var=Parameter(0.5, Positive()), nstarts=0, maxiter=1).fit(X[:50000], y[:50000])
F = slm.weights_.shape[0]
for name, fn in (("predict_moments", lambda: slm.predict_moments(X)), ("predict", lambda: slm.predict(X))):
    fn()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    ms = 1e3 * sorted(ts)[1]
    print("%s: F = %d, %d rows: %.1f ms%s" % (name, F, N, ms, "  = %.3f of 157.3 TF/s on F^2 + 2 d n flop per row" % ((F * F + 2.0 * d * n) * N / (ms * 1e-3) / 157.3e12) if name == "predict_moments" else ""))
