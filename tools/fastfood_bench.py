#!/usr/bin/env python3
"""FastFoodRBF f32 at config 4's shape, device-resident chunk of 262144 rows: ms per launch and TB/s of Phi written.
RR_FF_ABLATE (1 = no sincos, 2 = no global stores) and RR_FF_ROWS_PER_BLOCK are measurement switches of the launcher."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd.basis_functions import FastFoodRBF
N, d, nb = int(os.environ.get("FF_ROWS", 262144)), 128, 8192
X = np.random.RandomState(0).randn(N, d).astype(np.float32)
b = FastFoodRBF(nbases=nb, Xdim=d, random_state=1)
ff, _ = b._handles()
dev = ff.dev
F = 2 * ff.n
dX = dev.upload_matrix(X)
out = dev.malloc(N * F * 4)
best = 1e9
for rep in range(6):
    dev.timer_start()
    ff.transform_dev(dX, 1.0, out, np.float32)
    best = min(best, dev.timer_stop())
print("f32 chain ablate=%s: N=%d F=%d: %.3f ms, %.2f TB/s written (%.0f%% of 8 TB/s)" % (
    os.environ.get("RR_FF_ABLATE", "0"), N, F, best, N * F * 4 / best / 1e9, N * F * 4 / best / 1e9 / 8 * 100))
