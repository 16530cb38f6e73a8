"""Small-problem call latencies through the Python classes (typical revrand sizes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd.slm import StandardLinearModel
rs = np.random.RandomState(0)
for N, d, n in ((1000, 4, 64), (10000, 8, 256), (100000, 16, 512)):
    X = rs.randn(N, d); y = rs.randn(N)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=1)
    b.transform(X); b.grad(X); b.gram(X, y)
    def t(f, reps=20):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        return (time.perf_counter() - t0) / reps * 1e3
    print("N=%d d=%d F=%d: transform %.2f ms, grad %.2f ms, gram %.2f ms" % (
        N, d, 2 * n, t(lambda: b.transform(X)), t(lambda: b.grad(X)), t(lambda: b.gram(X, y))), flush=True)
    slm = StandardLinearModel(b, nstarts=0, maxiter=30)
    t0 = time.perf_counter(); slm.fit(X, y); dt = time.perf_counter() - t0
    print("   fit maxiter=30: %.3f s; predict_moments %.2f ms" % (dt, t(lambda: slm.predict_moments(X), 5)), flush=True)
