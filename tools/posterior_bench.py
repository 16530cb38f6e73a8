"""rr_posterior_dev (blocked Cholesky + inverse + reductions on the device) against the host solve_posdef."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip
from revrand_amd.linalg import solve_posdef
dev = _hip.get_device()
for F in (256, 512, 1024, 2048, 4096, 8192):
    rs = np.random.RandomState(0)
    A = rs.randn(F, 64)
    G = A @ A.T
    b = rs.randn(F)
    iL = np.ones(F)
    acc = dev.upload_vector(np.concatenate((G.ravel(), b)))
    dC = dev.malloc(F * F * 8)
    pG, pb = _hip.ctypes.c_void_p(acc.ptr.value), _hip.ctypes.c_void_p(acc.ptr.value + F * F * 8)
    dev.posterior(F, pG, pb, iL, 0.5, dC)
    t0 = time.perf_counter()
    for _ in range(3):
        m, dg, logdet, tr = dev.posterior(F, pG, pb, iL, 0.5, dC)
    dt = (time.perf_counter() - t0) / 3
    line = "F=%d: device posterior %.2f ms" % (F, dt * 1e3)
    if F <= 4096:
        t0 = time.perf_counter()
        Ch, ld = solve_posdef(np.diag(iL) + G / 0.5, np.eye(F))
        mh = Ch @ b / 0.5; trh = (G * Ch).sum()
        th = time.perf_counter() - t0
        line += ", host %.1f ms, |dlogdet| %.1e, m err %.1e" % (th * 1e3, abs(ld - logdet), np.abs(m - mh).max() / np.abs(mh).max())
    print(line, flush=True)
    acc.free(); dC.free()
