"""Host solve_posdef (Cholesky + inverse + logdet, as slm.py:155 uses it) at a few widths."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd.linalg import solve_posdef
for F in (512, 2048, 4096, 8192):
    rs = np.random.RandomState(0)
    A = rs.randn(F, 64)
    iC = A @ A.T / 64 + np.eye(F)
    t0 = time.perf_counter()
    C, ld = solve_posdef(iC, np.eye(F))
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    tr = (iC * C).sum(); m = C @ iC[0]
    dt2 = time.perf_counter() - t1
    print("F=%d: solve_posdef %.3f s, trace+gemv %.3f s, cores %d" % (F, dt, dt2, os.cpu_count()), flush=True)
