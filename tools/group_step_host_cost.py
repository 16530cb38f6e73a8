"""What a step of the resident SVI loop costs the HOST -- one context against a device group -- on a problem whose kernels take
next to nothing (N = 40 000, F = 32, minibatch 4096): the interval between the moments the loop queues its steps is then the
host's part of a step (cutting the minibatch, splitting it by owner, the per-member hand-over, the library call's launches).
RR_GLM_GROUP_TIMING=1: the library prints the host time inside rr_glm_sgd_group_step and how much of it was spent waiting for
step t - 2; RR_GLM_GROUP_THREADS=0: the members queued in turn by one thread.  `python tools/group_step_host_cost.py`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs  # noqa: E402
from revrand_amd import likelihoods as lk  # noqa: E402
from revrand_amd.btypes import Parameter, Positive  # noqa: E402
from revrand_amd.glm import GeneralizedLinearModel  # noqa: E402

N, d = 40000, 8
rs = np.random.RandomState(0)
X = rs.randn(N, d).astype(np.float32)
y = rs.poisson(np.exp(0.3 * X[:, 0])).astype(float)
for sampler in ("device", "host"):
    for devices in (None, [0], [0, 0], [0, 0, 0, 0], [0] * 8):
        for rep in range(2):
            g = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=16, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                                       K=2, nsamples=4, batch_size=2048 * max(1, len(devices or [0])), maxiter=300, nstarts=0, random_state=2,
                                       sampler=sampler, devices=devices)
            np.random.seed(1)
            g.fit(X, y)
        dt = 1e6 * np.diff(g.__dict__["_resident_clock"][20:-1])
        print("%-6s sampler, devices=%-12s: %6.0f us per step (median %6.0f)" % (sampler, devices, dt.mean(), np.median(dt)))
