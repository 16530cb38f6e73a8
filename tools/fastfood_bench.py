#!/usr/bin/env python3
"""Config 4 shape (FastFoodRBF nbases=8192, D=128 -> F=16384): the transform kernel device-resident (HIP events), and
the PCIe-inclusive host call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd.basis_functions import FastFoodRBF
N, d, nb = 262144, 128, 8192
X = np.random.RandomState(0).randn(N, d).astype(np.float32)
b = FastFoodRBF(nbases=nb, Xdim=d, random_state=1)
ff, rff = b._handles()
dev = ff.dev
F = 2 * ff.n
dX = dev.upload_matrix(X)
out = dev.malloc(N * F * 4)
for rep in range(3):
    dev.timer_start()
    ff.transform_dev(dX, 1.0, out)
    ms = dev.timer_stop()
print("device-resident: N=%d F=%d f32 out: %.2f ms, %.1f M rows/s, %.2f TB/s written (%.0f%% of 8 TB/s)" % (
    N, F, ms, N / ms / 1e3, N * F * 4 / ms / 1e9, N * F * 4 / ms / 1e9 / 8 * 100))
P = dev.download(out, (64, F), np.float32)
ref = ff.transform(X[:64], 1.0, out_dtype=np.float32)
print("device-resident vs host call, 64 rows: max abs diff %.1e" % float(np.abs(P - ref).max()))
out.free()
for rep in range(2):
    t0 = time.perf_counter()
    P = ff.transform(X[:65536], 1.0, out_dtype=np.float32)
    dt = time.perf_counter() - t0
    print("host call: N=%d F=%d  %.3f s (%.2f GB out, PCIe-inclusive %.0f rows/s)" % (65536, P.shape[1], dt, P.nbytes / 1e9, 65536 / dt))
