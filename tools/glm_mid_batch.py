"""The resident SVI loop (rr_glm_sgd, one step per library call) at MID-sized minibatches -- between the reference's default of
10 rows (the fused loop: tools/glm_fused_bench.py) and config 5's 65 536: the interval between queued steps for a few shapes.
`python tools/glm_mid_batch.py` (under rocprofv3 --kernel-trace --stats for the kernels of a step)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs  # noqa: E402
from revrand_amd import likelihoods as lk  # noqa: E402
from revrand_amd.btypes import Parameter, Positive  # noqa: E402
from revrand_amd.glm import GeneralizedLinearModel  # noqa: E402

rs = np.random.RandomState(0)
N, d = 400_000, 16
X = rs.randn(N, d).astype(np.float32)
y = rs.poisson(np.exp(0.3 * X[:, 0])).astype(float)
shapes = [(int(v) for v in s.split(",")) for s in os.environ.get("SHAPES", "1024,256;4096,256;4096,1024;16384,512;16384,1024").split(";")]
for M, n in shapes:
    for rep in range(2):
        g = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                                   K=10, nsamples=50, batch_size=M, maxiter=150, nstarts=0, random_state=2, sampler="device")
        np.random.seed(1)
        g.fit(X, y)
    dt = 1e6 * np.diff(g.__dict__["_resident_clock"][20:-1])
    flops = 3 * 2.0 * 500 * M * 2 * n
    print("minibatch %6d, F = %4d: %7.0f us per step (median %7.0f) = %.3f of the f32 MFMA peak on the step's three products"
          % (M, 2 * n, dt.mean(), np.median(dt), flops / (np.median(dt) * 1e-6) / 157.3e12))
