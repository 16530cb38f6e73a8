"""Where a 41 500-value rr_legacy_randn call spends its time on this host: the C call alone against the Python wrapper
(get_state / set_state of the RandomState), for 1 / 2 / 4 / 8 worker threads."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from revrand_amd import _hip  # noqa: E402

lib = _hip.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 41500
rs = np.random.RandomState(1)
out = np.empty(n, dtype=np.float32)
st = rs.get_state()
key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
pos, has, g = ctypes.c_int32(int(st[2])), ctypes.c_int32(0), ctypes.c_double(0.0)
for threads in (1, 2, 4, 8):
    for _ in range(20):
        lib.rr_legacy_randn(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), ctypes.byref(has), ctypes.byref(g),
                            out.ctypes.data_as(ctypes.c_void_p), 0, n, threads)
    t0 = time.perf_counter()
    for _ in range(200):
        lib.rr_legacy_randn(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), ctypes.byref(has), ctypes.byref(g),
                            out.ctypes.data_as(ctypes.c_void_p), 0, n, threads)
    c_us = (time.perf_counter() - t0) / 200 * 1e6
    os.environ["RR_RANDN_THREADS"] = str(threads)
    t0 = time.perf_counter()
    for _ in range(200):
        _hip.legacy_randn(rs, n, np.float32, out=out)
    w_us = (time.perf_counter() - t0) / 200 * 1e6
    print("threads %d: C call %.1f us, through the wrapper %.1f us" % (threads, c_us, w_us))
