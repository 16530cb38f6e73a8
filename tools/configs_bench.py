#!/usr/bin/env python3
"""BASELINE.json's five configurations on one MI355X, end to end through the estimator / basis classes
(one GPU's share for the 8-GPU config 3).  Prints one line per config; numbers go into DESIGN.md section 8."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd import likelihoods as lk
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.glm import GeneralizedLinearModel
from revrand_amd.slm import StandardLinearModel

which = set(sys.argv[1:]) or {"1", "2", "3", "4", "5"}
rng = np.random.default_rng(0)


def gaussian_data(N, d):
    X = rng.standard_normal((N, d), dtype=np.float32)
    w = rng.standard_normal(d, dtype=np.float32)
    y = np.sin(X @ w / np.sqrt(d)).astype(np.float32) + 0.1 * rng.standard_normal(N, dtype=np.float32)
    return X, y


def timed_elbo(slm, X, y, var, reg, hyp, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        slm._elbo(X, y, var, reg, hyp)
        ts.append(time.perf_counter() - t0)
    return min(ts)


out = {}
if "1" in which:  # C1: fit, RandomRBF nbases=256, D=8, N=10k
    X, y = gaussian_data(10_000, 8)
    slm = StandardLinearModel(bs.RandomRBF(nbases=256, Xdim=8, random_state=1), nstarts=0, maxiter=50)
    slm.fit(X, y)  # warm
    t0 = time.perf_counter()
    slm = StandardLinearModel(bs.RandomRBF(nbases=256, Xdim=8, random_state=1), nstarts=0, maxiter=50).fit(X, y)
    out["C1 fit (N=10k, D=8, F=512, maxiter=50)"] = "%.2f s" % (time.perf_counter() - t0)
    print(out, flush=True)

if "2" in which:  # C2: RandomRBF F=4096, D=32, N=1M
    X, y = gaussian_data(1_000_000, 32)
    b = bs.RandomRBF(nbases=2048, Xdim=32, random_state=1, lenscale=Parameter(np.ones(32), Positive()))
    st = b.device_fit_state(X, y)
    st.gram_device(np.ones(32))
    t0 = time.perf_counter(); st.gram_device(np.ones(32)); dt = time.perf_counter() - t0
    out["C2 Phi + Gram (N=1M, D=32, F=4096), resident"] = "%.3f s = %.2f M rows/s" % (dt, 1.0 / dt)
    slm = StandardLinearModel(b); slm.obj_ = -np.inf; slm._state = st
    out["C2 one _elbo (ARD, device posterior)"] = "%.3f s" % timed_elbo(slm, X, y, 0.5, 1.0, np.ones(32))
    st.release(); slm._state = None
    t0 = time.perf_counter()
    slm = StandardLinearModel(b, nstarts=0, maxiter=10).fit(X, y)
    out["C2 fit maxiter=10"] = "%.1f s" % (time.perf_counter() - t0)
    print(out, flush=True)
    del X, y

if "3" in which:  # C3: RandomMatern52 n=4096 + LinearBasis, D=64, one GPU's share of N=10M / 8
    N = 1_250_000
    X, y = gaussian_data(N, 64)
    cat = bs.RandomMatern52(nbases=4096, Xdim=64, random_state=1, lenscale=Parameter(np.ones(64), Positive())) \
        + bs.LinearBasis(onescol=True)
    st = cat.device_fit_state(X, y)
    st.gram_device([np.ones(64)])
    t0 = time.perf_counter(); st.gram_device([np.ones(64)]); dt = time.perf_counter() - t0
    out["C3 concat Gram (N=1.25M = 10M/8, D=64, F=8257), resident"] = "%.3f s = %.2f M rows/s" % (dt, N / dt / 1e6)
    slm = StandardLinearModel(cat); slm.obj_ = -np.inf; slm._state = st
    out["C3 one _elbo"] = "%.3f s" % timed_elbo(slm, X, y, 0.5, [1.0, 1.0], np.ones(64), reps=2)
    st.release(); slm._state = None
    print(out, flush=True)
    del X, y

if "4" in which:  # C4: FastFoodRBF F=16384, D=128, N=4M (Phi streamed to the host in float32 would be 262 GB: time the
    # device side through the resident Gram of a 1M-row slice and the host transform of 100k rows)
    X, _ = gaussian_data(200_000, 128)
    f = bs.FastFoodRBF(nbases=8192, Xdim=128, random_state=1)
    f.transform(X[:1000])
    t0 = time.perf_counter(); P = f.transform(X[:100_000]); dt = time.perf_counter() - t0
    out["C4 FastFood transform to host float64 (N=100k, F=16384)"] = "%.3f s = %.1f GB/s" % (dt, P.nbytes / dt / 1e9)
    del P
    print(out, flush=True)

if "5" in which:  # C5: GLM Poisson, RandomRBF F=2048, N=2M, minibatch 65536
    N, d = 2_000_000, 32
    X = rng.standard_normal((N, d), dtype=np.float32).astype(np.float64)
    rate = np.exp(0.6 * np.sin(X[:, 0]) + 0.3 * X[:, 1])
    y = rng.poisson(rate).astype(np.float64)
    basis = bs.RandomRBF(nbases=1024, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    for sampler in ("host", "device"):
        glm = GeneralizedLinearModel(lk.Poisson(), basis, K=10, nsamples=50, batch_size=65536, maxiter=60, nstarts=4,
                                     random_state=2, sampler=sampler)
        t0 = time.perf_counter(); glm.fit(X, y); dt = time.perf_counter() - t0
        out["C5 GLM fit (N=2M, F=2048, K=10, L=50, batch 65536, 64 SVI steps), %s sampler" % sampler] = \
            "%.1f s = %.0f ms/step" % (dt, dt / 64 * 1e3)
    print(out, flush=True)
print("RESULT " + json.dumps(out))
