"""the opt-in deterministic reduction (rr_set_deterministic / RR_DETERMINISTIC=1).  The reference's
`Phi.T.dot(Phi)` (slm.py:146) gives the same bits every run; by default the kernels here sum with floating-point atomics
(last bits vary).  In deterministic mode `basis.gram` and a whole `_elbo` are bitwise reproducible, and two ranks holding
the same reduced statistics agree bit for bit without the broadcast of the results."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import revrand_oracle as orc
from conftest import ROOT, normwise

pytestmark = pytest.mark.gpu


def _setup():
    import revrand_amd.basis_functions as bs
    from revrand_amd import _hip
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    return bs, _hip, Parameter, Positive, StandardLinearModel


def _elbo_once(SLM, basis, X, y, var, reg, ls):
    slm = SLM(basis)
    slm.obj_ = -np.inf
    slm._state = slm._make_state(X, y)
    assert slm._state is not None
    f, (gv, gr, gh) = slm._elbo(X, y, var, reg, ls)
    C = slm._state.best_covariance() if getattr(slm._state, "best_on_device", False) else slm.covariance_
    slm._state.release()
    slm._state = None
    return np.concatenate(([f, gv], np.atleast_1d(gr), np.atleast_1d(gh), slm.weights_)), np.array(C)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_gram_and_elbo_are_bitwise_reproducible(dtype):
    bs, _hip, Parameter, Positive, SLM = _setup()
    os.environ["RR_POSDEF"] = "device"
    dev = _hip.get_device()
    rs = np.random.RandomState(0)
    N, d, n = (150_000, 8, 300) if dtype == "f32" else (60_000, 8, 200)   # F = 600 / 400: several K-splits, a ragged block
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)).astype(np.float32)
    ls = np.linspace(0.8, 1.3, d)
    prev = dev.set_deterministic(True)
    try:
        assert dev.deterministic
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()), dtype=dtype)
        runs = [basis.gram(X, y, ls) for _ in range(3)]
        for G, b, yty in runs[1:]:
            assert np.array_equal(G, runs[0][0]) and np.array_equal(b, runs[0][1]) and yty == runs[0][2]
        assert np.array_equal(runs[0][0], runs[0][0].T)
        e = [_elbo_once(SLM, basis, X, y, 0.3, 1.2, ls) for _ in range(3)]
        for v, C in e[1:]:
            assert np.array_equal(v, e[0][0]) and np.array_equal(C, e[0][1])
        # a concatenation (feature matrix, rider column, ragged SYRK tiles) -- the f32 route only
        if dtype == "f32":
            cat = bs.RandomMatern52(nbases=n, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive())) \
                + bs.LinearBasis(onescol=True)
            c = [_elbo_once(SLM, cat, X, y, 0.3, [1.2, 0.8], ls) for _ in range(2)]
            assert np.array_equal(c[0][0], c[1][0]) and np.array_equal(c[0][1], c[1][1])
            Gc = [cat.gram(X, y, ls) for _ in range(2)]
            assert np.array_equal(Gc[0][0], Gc[1][0]) and np.array_equal(Gc[0][1], Gc[1][1])
            # host feature matrices through rr_dense_gram, and the contraction entry point
            Phi = basis.transform(X[:20000], ls)
            D1, D2 = _hip.dense_gram(Phi, y[:20000].astype(np.float64)), _hip.dense_gram(Phi, y[:20000].astype(np.float64))
            assert np.array_equal(D1[0], D2[0]) and np.array_equal(D1[1], D2[1]) and D1[2] == D2[2]
            E = rs.randn(20000, 2 * n)
            assert np.array_equal(basis.grad_contract(X[:20000], E, ls), basis.grad_contract(X[:20000], E, ls))
        det_G, det_v = runs[0][0], e[0][0]
    finally:
        dev.set_deterministic(prev)
        os.environ.pop("RR_POSDEF", None)
    # the same numbers as the default (atomic) mode to rounding, and as the oracle to the path's tolerance
    G, b, yty = basis.gram(X, y, ls)
    tol = 1e-6 if dtype == "f32" else 1e-13
    assert normwise(det_G, G) < tol
    os.environ["RR_POSDEF"] = "device"
    try:
        v, _ = _elbo_once(SLM, basis, X, y, 0.3, 1.2, ls)
    finally:
        os.environ.pop("RR_POSDEF", None)
    assert normwise(det_v, v) < (1e-4 if dtype == "f32" else 1e-9)
    ns = 4000
    Gs, _, _ = basis.gram(X[:ns], y[:ns], ls)
    prev = dev.set_deterministic(True)
    try:
        Gd, bd, _ = basis.gram(X[:ns], y[:ns], ls)
    finally:
        dev.set_deterministic(prev)
    Go, bo, _ = orc.gram_stats(orc.rff_transform(X[:ns].astype(np.float64), basis.W, ls), y[:ns].astype(np.float64))
    assert normwise(Gd, Go) < (1e-5 if dtype == "f32" else 1e-12) and normwise(bd, bo) < (1e-5 if dtype == "f32" else 1e-12)


def test_split_engines_are_refused_in_deterministic_mode():
    bs, _hip, Parameter, Positive, SLM = _setup()
    dev = _hip.get_device()
    basis = bs.RandomRBF(nbases=64, Xdim=4, random_state=1)
    X = np.random.RandomState(0).randn(500, 4).astype(np.float32)
    prev, eng = dev.set_deterministic(True), dev.set_gram_engine("fp16x3")
    try:
        with pytest.raises(_hip.HipError, match="deterministic"):
            basis.gram(X, None, 1.0)
    finally:
        dev.set_gram_engine(eng)
        dev.set_deterministic(prev)


_TWO_RANKS_DET = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
os.environ["RR_POSDEF"] = "device"
from revrand_amd import _hip, parallel
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel as SLM
comm = parallel.init_rccl_from_env()
parallel.set_comm(comm)
assert _hip.get_device().deterministic            # RR_DETERMINISTIC=1 in the environment
rank, world = comm.rank, comm.world
casts = []
orig = comm.broadcast_host
comm.broadcast_host = lambda arr, root=0: (casts.append(np.size(arr)), orig(arr, root))[1]
rs = np.random.RandomState(0)
N, d, n = 40001, 6, 160
X = rs.randn(N, d).astype(np.float32); y = (np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)).astype(np.float32)
a, b = parallel.shard_bounds(N, rank, world)
basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
slm = SLM(basis, var=Parameter(0.1, Positive()), nstarts=0, maxiter=6, distributed=True, random_state=0)
slm.obj_ = -np.inf
slm._state = slm._make_state(X[a:b], y[a:b])
out = {"rank": rank, "elbo": []}
for rep in range(2):
    f, (gv, gr, gh) = slm._elbo(X[a:b], y[a:b], 0.3, 1.2, np.linspace(0.8, 1.3, d))
    out["elbo"].append([float(f), float(gv), float(gr)] + np.asarray(gh).tolist())
fo = slm._elbo_objective(X[a:b], y[a:b], 0.3, 1.2, np.linspace(0.8, 1.3, d))
out["objective_only"] = float(fo)
slm._state.release(); slm._state = None
slm.fit(X[a:b], y[a:b])
out["fit"] = [float(slm.var_), float(slm.regularizer_)] + np.asarray(slm.hypers_).tolist() + [float(slm.obj_)] + slm.weights_.tolist()
out["cov_sum"] = float(np.asarray(slm.covariance_).sum())
out["broadcasts"] = casts
comm.barrier()
comm.close()
sys.stdout.flush(); print("\nRESULT" + json.dumps(out) + "ENDRESULT", flush=True)
'''


def test_two_rccl_ranks_agree_bitwise_without_the_result_broadcast(tmp_path):
    from test_gpu_comm import _run_ranks
    res = sorted(_run_ranks(_TWO_RANKS_DET % ROOT, 2, tmp_path, extra_env={"RR_DETERMINISTIC": "1"}), key=lambda r: r["rank"])
    assert [r["rank"] for r in res] == [0, 1]
    for r in res:
        assert r["broadcasts"] == [], r["broadcasts"]          # two exchanges per evaluation, nothing else
        assert r["elbo"][0] == r["elbo"][1]                       # reproducible within a process
    assert res[0]["elbo"] == res[1]["elbo"] and res[0]["objective_only"] == res[1]["objective_only"]
    assert res[0]["fit"] == res[1]["fit"] and res[0]["cov_sum"] == res[1]["cov_sum"]
    assert abs(res[0]["objective_only"] - res[0]["elbo"][0][0]) < 1e-6 * abs(res[0]["elbo"][0][0])
