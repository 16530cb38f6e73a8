#!/usr/bin/env python3
"""Xdim > 128 path (phases by GEMM + trig kernel), device-resident: Gram and transform at d = 256 / 784."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip

def run(N, d, n, compute="f32"):
    W = np.random.RandomState(42).randn(d, n)
    h = _hip.RffHandle(W, compute=compute)
    dev = h.dev
    rng = np.random.default_rng(0)
    X = (rng.standard_normal((N, d), dtype=np.float32) / np.sqrt(d / 8)).astype(np.float32)
    y = rng.standard_normal(N, dtype=np.float32)
    dX = h.upload(X); dy = dev.upload_vector(y)
    F = 2 * n
    acc = dev.zeros((F * F + F + 1) * 8); base = acc.ptr.value
    def gram():
        h.gram_dev(dX, dy, 1.0, ctypes.c_void_p(base), ctypes.c_void_p(base + F * F * 8), ctypes.c_void_p(base + (F * F + F) * 8))
    gram(); dev.sync()
    for _ in range(2):
        gram(); f, g, dg, k = h.gram_timings()
    print("gram %s N=%d d=%d F=%d: features (transpose + GEMM + trig) %.2f ms, syrk %.2f ms -> %.2f M rows/s; features alone %.1f M rows/s, "
          "%.2f TB/s of P written" % (compute, N, d, F, f, g, N / (f + g) / 1e3, N / f / 1e3, N * F * (4 if compute == "f32" else 8) / f / 1e9))
    out = dev.malloc(N * F * 4)
    ls = np.ones(1)
    def tr():
        _hip._check(dev.lib, dev.lib.rr_rff_transform_dev(h.h, dX.ptr, 0, N, dX.ld, ls.ctypes.data_as(ctypes.c_void_p), 1,
                                                          out.ptr, _hip.rr_dtype(np.float32), F))
    tr(); dev.sync()
    dev.timer_start()
    for _ in range(3): tr()
    ms = dev.timer_stop() / 3
    print("transform %s -> f32 N=%d d=%d F=%d: %.2f ms, %.1f M rows/s, %.2f TB/s written" % (compute, N, d, F, ms, N / ms / 1e3, N * F * 4 / ms / 1e9))
    out.free(); dX.free(); dy.free(); acc.free()

run(1_000_000, 256, 1024)
run(500_000, 784, 2048)
run(100_000, 256, 1024, "f64")
