"""Host-buffer (PCIe-inclusive) calls: transform to a float64 array, Gram from host X, FastFood transform."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd.basis_functions import RandomRBF, FastFoodRBF
rs = np.random.RandomState(0)
N, d, n = 200_000, 32, 2048
X = rs.randn(N, d)
y = rs.randn(N)
b = RandomRBF(nbases=n, Xdim=d, random_state=1)
b.transform(X[:1000])
for rep in range(2):
    t0 = time.perf_counter(); P = b.transform(X); dt = time.perf_counter() - t0
    print("transform N=%d F=%d -> float64 host array: %.3f s, %.2f GB/s of Phi, %.0f rows/s" % (N, 2 * n, dt, P.nbytes / dt / 1e9, N / dt), flush=True)
del P
for rep in range(2):
    t0 = time.perf_counter(); G, bb, yty = b.gram(X, y); dt = time.perf_counter() - t0
    print("gram from host X N=%d: %.3f s, %.0f rows/s" % (N, dt, N / dt), flush=True)
f = FastFoodRBF(nbases=8192, Xdim=128, random_state=1)
Xf = rs.randn(40_000, 128)
f.transform(Xf[:100])
for rep in range(2):
    t0 = time.perf_counter(); P = f.transform(Xf); dt = time.perf_counter() - t0
    print("fastfood transform N=%d F=%d -> float64: %.3f s, %.2f GB/s" % (len(Xf), P.shape[1], dt, P.nbytes / dt / 1e9), flush=True)
