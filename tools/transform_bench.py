#!/usr/bin/env python3
"""Random Fourier `transform` device-resident (rr_rff_transform_dev) at the headline shape: TB/s of Phi written, for
f32 and f64 arithmetic (X and Phi in the arithmetic's dtype)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip
from revrand_amd.basis_functions import RandomRBF
N, d, nb = int(os.environ.get("TB_ROWS", 500000)), int(os.environ.get("TB_DIM", 32)), int(os.environ.get("TB_NBASES", 2048))
for dtype, npdt in (("f32", np.float32), ("f64", np.float64)):
    X = np.random.RandomState(0).randn(N, d).astype(npdt)
    b = RandomRBF(nbases=nb, Xdim=d, random_state=1, dtype=dtype)
    h = b._handle()
    dev = h.dev
    F = 2 * h.n
    dX = dev.upload_matrix(X, ld_dev=max(h.padded_dim, d))
    out = dev.malloc(N * F * X.itemsize)
    ls = np.array([1.3])
    best = 1e9
    for rep in range(5):
        dev.timer_start()
        _hip._check(dev.lib, dev.lib.rr_rff_transform_dev(h.h, dX.ptr, _hip.rr_dtype(X.dtype), N, dX.ld,
                                                          ls.ctypes.data_as(ctypes.c_void_p), 1, out.ptr,
                                                          _hip.rr_dtype(X.dtype), F))
        best = min(best, dev.timer_stop())
    print("%s transform: N=%d d=%d F=%d: %.3f ms, %.1f M rows/s, %.2f TB/s written (%.0f%% of 8 TB/s)" % (
        dtype, N, d, F, best, N / best / 1e3, N * F * X.itemsize / best / 1e9, N * F * X.itemsize / best / 1e9 / 8 * 100))
