#!/usr/bin/env python3
"""NumPy's legacy randn against rr_legacy_randn (same stream, bit for bit) at config 5's per-step size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip
n = 10 * 50 * 2048
a, b = np.random.RandomState(0), np.random.RandomState(0)
for thr in (1, 2, 4, 8, 16):
    best = 1e9
    for rep in range(5):
        t = time.perf_counter(); x = _hip.legacy_randn(b, n, np.float32, threads=thr); best = min(best, time.perf_counter() - t)
    print("rr_legacy_randn %2d worker threads: %.2f ms" % (thr, best * 1e3))
best = 1e9
for rep in range(5):
    t = time.perf_counter(); y = a.randn(500, 2048).astype(np.float32); best = min(best, time.perf_counter() - t)
print("numpy RandomState.randn: %.2f ms" % (best * 1e3))
