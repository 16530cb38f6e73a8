#!/usr/bin/env python3
"""Config 4 shape (FastFoodRBF nbases=8192, D=128 -> F=16384): kernel time of the transform, read from
rocprofv3's kernel trace (run under tools/prof.sh-style rocprofv3 --kernel-trace --stats)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd.basis_functions import FastFoodRBF
N, d, nb = 65536, 128, 8192
X = np.random.RandomState(0).randn(N, d).astype(np.float32)
b = FastFoodRBF(nbases=nb, Xdim=d, random_state=1)
h = b._handles()[0]
for rep in range(3):
    t0 = time.perf_counter()
    P = h.transform(X, 1.0, out_dtype=np.float32)
    dt = time.perf_counter() - t0
    print("host call: N=%d F=%d  %.3f s (%.2f GB out, PCIe-inclusive %.0f rows/s)" % (N, P.shape[1], dt, P.nbytes / 1e9, N / dt))
print("bytes per row (4d + 4F):", 4 * d + 4 * P.shape[1])
