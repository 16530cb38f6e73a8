"""Gram pass (features + Phi^T Phi + Phi^T y, resident X) at small feature counts: rr_syrk_f32_small_kernel (128 x 128 tiles)
against the 256 x 256 tile kernels (RR_SYRK_SMALL=0) -- where the switch between them belongs.  ms per pass, device timer."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip  # noqa: E402

dev = _hip.get_device()
for n, N in ((256, 10_000), (256, 50_000), (256, 262_144), (512, 10_000), (512, 44_484), (512, 131_072), (512, 262_144), (512, 1_000_000)):
    d = 8
    rs = np.random.RandomState(0)
    W = rs.randn(d, n)
    X = rs.randn(N, d).astype(np.float32)
    y = rs.randn(N).astype(np.float32)
    h = _hip.RffHandle(W, compute="f32")
    dX, dy = h.upload(X), dev.upload_vector(y)
    F = 2 * n
    acc = dev.zeros((F * F + F + 1) * 8)
    base = acc.ptr.value
    ptrs = [_hip.ctypes.c_void_p(base + o * 8) for o in (0, F * F, F * F + F)]

    def step():
        dev.memset(acc)
        h.gram_dev(dX, dy, 1.0, *ptrs)
    for _ in range(3):
        step()
    ts = []
    for _ in range(7):
        dev.sync()
        dev.timer_start()
        step()
        ts.append(dev.timer_stop())
    ms = float(np.median(ts))
    print("F=%d N=%d: %.3f ms  (%.2f of the f32 MFMA peak on F(F+1) flop/row)  kernel %s"
          % (F, N, ms, F * (F + 1.0) * N / (ms * 1e-3) / 157.3e12, h.gram_kernel_name()))
    for b in (dX, dy, acc):
        b.free()
