#!/usr/bin/env python3
"""FastFoodRBF(dtype="f64") at config 4's shape, device-resident: the float64 chain kernel (f64 in, f64 out)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd.basis_functions import FastFoodRBF
N, d, nb = 65536, 128, 8192
X = np.random.RandomState(0).randn(N, d)
b = FastFoodRBF(nbases=nb, Xdim=d, random_state=1, dtype="f64")
ff, _ = b._handles()
dev = ff.dev
F = 2 * ff.n
dX = dev.upload_matrix(X)
out = dev.malloc(N * F * 8)
for rep in range(3):
    dev.timer_start()
    ff.transform_dev(dX, 1.0, out, np.float64)
    ms = dev.timer_stop()
print("f64 chain: N=%d F=%d f64 out: %.2f ms, %.1f M rows/s, %.2f TB/s written (%.0f%% of 8 TB/s)" % (
    N, F, ms, N / ms / 1e3, N * F * 8 / ms / 1e9, N * F * 8 / ms / 1e9 / 8 * 100))
