#!/usr/bin/env python3
"""Config 4 shape (FastFoodRBF nbases=8192, D=128 -> F=16384): the FWHT-chain kernel against the dense equivalent
W = _makeVX(I) through the MFMA feature kernel, both device-resident and HIP-event timed."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip
from revrand_amd.basis_functions import FastFoodRBF
N, d, nb = 131072, 128, 8192
X = np.random.RandomState(0).randn(N, d).astype(np.float32)
b = FastFoodRBF(nbases=nb, Xdim=d, random_state=1)
ff, rff = b._handles()
dev = rff.dev
F = 2 * rff.n
dX = rff.upload(X)
out = dev.malloc(N * F * 4)
ls = np.array([1.0])
for name in ("dense MFMA",):
    for rep in range(3):
        dev.timer_start()
        _hip._check(dev.lib, dev.lib.rr_rff_transform_dev(rff.h, dX.ptr, 0, N, dX.ld, ls.ctypes.data_as(ctypes.c_void_p), 1,
                                                          out.ptr, 0, F))
        ms = dev.timer_stop()
    print("%s: N=%d F=%d %.2f ms, %.1f M rows/s, %.2f TB/s written (%.0f%% of 8 TB/s)" % (
        name, N, F, ms, N / ms / 1e3, N * F * 4 / ms / 1e9, N * F * 4 / ms / 1e9 / 8 * 100))
P = dev.download(out, (256, F), np.float32)
ref = ff.transform(X[:256], 1.0, out_dtype=np.float32)
print("max |dense - chain| on 256 rows:", float(np.abs(P - ref).max()))
