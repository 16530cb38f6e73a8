#!/usr/bin/env python3
"""Prediction calls of a fitted StandardLinearModel at F ~ 4096: predict (mean only), predict_moments (triangular C),
single basis and concatenation, large batch and small-batch latency."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd import StandardLinearModel
from revrand_amd.btypes import Parameter, Positive
rng = np.random.default_rng(0)
N, d = 300000, 32
X = rng.standard_normal((N, d), dtype=np.float32); y = np.sin(X[:, 0]).astype(np.float32)
for name, basis in (("RandomRBF F=4096", bs.RandomRBF(nbases=2048, Xdim=d, random_state=1, lenscale=Parameter(1.0, Positive()),
                                                      regularizer=Parameter(1.0, Positive()))),
                    ("RandomMatern52 + LinearBasis F=4129", bs.RandomMatern52(nbases=2048, Xdim=d, random_state=1,
                                                                               lenscale=Parameter(1.0, Positive()),
                                                                               regularizer=Parameter(1.0, Positive()))
                     + bs.LinearBasis(onescol=True, regularizer=Parameter(1.0, Positive())))):
    slm = StandardLinearModel(basis, var=Parameter(0.5, Positive()), nstarts=0, maxiter=1).fit(X[:50000], y[:50000])
    for rows in (64, N):
        for f in ("predict", "predict_moments"):
            getattr(slm, f)(X[:rows])
            reps = 5 if rows < 1000 else 1
            t0 = time.perf_counter()
            for _ in range(reps):
                getattr(slm, f)(X[:rows])
            print("%-38s rows=%-7d %-16s %.2f ms" % (name, rows, f, (time.perf_counter() - t0) / reps * 1e3), flush=True)
