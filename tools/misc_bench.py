#!/usr/bin/env python3
"""Secondary kernels, device-resident, HIP-event timed: transform (f32/f64), f64 Gram."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip

def transform_bench(compute, out_np, N=500_000, d=32, n=2048):
    W = np.random.RandomState(42).randn(d, n)
    h = _hip.RffHandle(W, compute=compute)
    dev = h.dev
    X = np.random.default_rng(0).standard_normal((N, d), dtype=np.float32)
    dX = h.upload(X)
    F = 2 * n
    out = dev.malloc(N * F * np.dtype(out_np).itemsize)
    ls = np.ones(1)
    def run():
        _hip._check(dev.lib, dev.lib.rr_rff_transform_dev(h.h, dX.ptr, 0, N, dX.ld, ls.ctypes.data_as(ctypes.c_void_p), 1,
                                                          out.ptr, _hip.rr_dtype(out_np), F))
    run(); dev.sync()
    dev.timer_start()
    for _ in range(3): run()
    ms = dev.timer_stop() / 3
    by = N * F * np.dtype(out_np).itemsize
    print("transform compute=%s out=%s N=%d F=%d: %.2f ms, %.1f M rows/s, %.2f TB/s written (%.0f%% of 8 TB/s)" % (
        compute, np.dtype(out_np).name, N, F, ms, N / ms / 1e3, by / ms / 1e9, 100 * by / ms / 1e9 / 8))
    out.free(); dX.free()

def gram64_bench(N=200_000, d=32, n=2048):
    W = np.random.RandomState(42).randn(d, n)
    h = _hip.RffHandle(W, compute="f64")
    dev = h.dev
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N, d)); y = rng.standard_normal(N)
    dX = h.upload(X); dy = dev.upload_vector(y)
    F = 2 * n
    acc = dev.zeros((F * F + F + 1) * 8); base = acc.ptr.value
    def run():
        h.gram_dev(dX, dy, 1.0, ctypes.c_void_p(base), ctypes.c_void_p(base + F * F * 8), ctypes.c_void_p(base + (F * F + F) * 8))
    run(); dev.sync()
    for _ in range(2):
        run(); f, g, dg, k = h.gram_timings()
    fl = F * (F + 1.0) * N
    print("f64 gram N=%d F=%d: features %.2f ms, syrk_f64 %.2f ms -> %.1f TFLOP/s algorithmic = %.0f%% of the 78.6 TF f64 MFMA peak; %.2f M rows/s" % (
        N, F, f, g, fl / g / 1e9, 100 * fl / g / 1e9 / 78.6, N / (f + g) / 1e3))

transform_bench("f32", np.float32)
transform_bench("f32", np.float64)
transform_bench("f64", np.float64, N=200_000)
gram64_bench()
