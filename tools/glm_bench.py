"""Config 5 (BASELINE.json configs[4]) shape: GLM Poisson, RandomRBF n=1024 (F=2048), d=32 ARD, K=10, L=50,
minibatch M rows: time of one `_elbo` SVI step on the device, and the same step through the NumPy oracle on
the host for a bounded sample.  Prints one JSON line (minibatch-rows/s)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--nbases", type=int, default=1024)
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--L", type=int, default=50)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu-rows", type=int, default=2048)
    a = ap.parse_args()
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    rs = np.random.RandomState(0)
    M, d, n, K, L = a.rows, a.dim, a.nbases, a.K, a.L
    X = rs.randn(M, d).astype(np.float32)
    y = rs.poisson(np.exp(0.5 * np.sin(X[:, 0]))).astype(float)
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    glm = GeneralizedLinearModel(lk.Poisson(), basis, K=K, nsamples=L, random_state=2)
    glm.B_, glm.D_ = 30.0, 2 * n
    glm._GeneralizedLinearModel__it = 1   # a plain SGD iteration (no ELBO logging)
    m = 0.1 * rs.randn(2 * n, K)
    C = rs.gamma(2., 0.5, size=(2 * n, K))
    ls = np.linspace(0.8, 1.5, d)
    glm._elbo(m, C, 1.0, [], ls, X, y)   # warm-up (allocations)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        glm._elbo(m, C, 1.0, [], ls, X, y)
    dt = (time.perf_counter() - t0) / a.steps
    # pieces: device step only
    feats = glm._features()
    feats.assemble(X, [ls])
    WS = rs.randn(K * L, 2 * n)
    dev = feats.dev
    dev.sync()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        feats.glm_step(y, None, lk.RR_LIK_POISSON_EXP, 0.0, WS, K, L)
        feats.glm_basis_grads(X)
    dt_dev = (time.perf_counter() - t1) / a.steps
    glm._release_features()
    out = {"metric": "GLM SVI minibatch-rows/s (config 5 shape)", "rows": M, "d": d, "F": 2 * n, "K": K, "L": L,
           "elbo_step_ms": dt * 1e3, "rows_per_s": M / dt, "device_step_plus_grads_ms": dt_dev * 1e3,
           "gemm_flops_per_step": 3 * 2.0 * K * L * M * 2 * n,
           "gemm_tflops_at_step_time": 3 * 2.0 * K * L * M * 2 * n / dt / 1e12}
    if a.cpu_rows:
        import revrand_oracle as orc
        Mc = min(a.cpu_rows, M)
        Xc, yc = X[:Mc].astype(float), y[:Mc]
        e = rs.randn(K, L, 2 * n)
        t2 = time.perf_counter()
        Phi = orc.rff_transform(Xc, basis.W, ls)
        dP = orc.rff_grad(Xc, basis.W, ls)
        orc.glm_elbo(m, C, np.ones(2 * n), slice(None), "poisson_exp", [], (), Phi, [dP[:, :, i] for i in range(d)], yc, e, 30.0)
        tc = time.perf_counter() - t2
        out["cpu_port"] = {"rows": Mc, "seconds": tc, "rows_per_s": Mc / tc, "threads": os.cpu_count()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
