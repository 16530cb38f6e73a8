"""rr_posterior_dev (`solve_posdef` of mathfun/linalg.py:84-125 on the device, plus m, diag C, log|iC| and
sum(G o C)) at the sizes its panel pipeline was written for -- F = 4096 (32 panels), 8257 (BASELINE config 3: 65 panels,
ragged last one) and 16384 (config 4's width, 128 panels) -- against the ORACLE's solve_posdef, in the default (f64
atomics) and the deterministic reduction mode; and its failure path: a matrix that stops being positive definite in a LATE
panel, with the look-ahead streams in flight, must come back as RR_ERR_NOT_POSDEF (`None`) with every stream drained, like
the reference's `if np.any(U.diagonal() < CHOLTHRESH)` (linalg.py:110-123) before its SVD route."""
import os

import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise

pytestmark = pytest.mark.gpu

# The largest size is config 4's full width, F = 16384 (128 panels: the paired-panel schedule over most of them).  Up to
# F_FULL_INVERSE the oracle forms the whole inverse; above it -- where the host's inverse alone takes the better part of a
# minute -- the oracle's solve_posdef is asked for 128 random COLUMNS of the inverse and for the mean (one Cholesky and 129
# right-hand sides: still the oracle, still every panel of the device's factor and substitution behind each column), plus
# log|iC|; sum(G o C) is then held to the host sum over the device's own C.  RR_TEST_FULL=1: the whole inverse at every size.
F_LARGE = 16384
F_FULL_INVERSE = 1 << 30 if os.environ.get("RR_TEST_FULL") == "1" else 12288
NCOLS = 128

_CASES = {}


def _case(F):
    """(G, b, iL, var) with G the Gram of F / 8 random rows (as a fit with fewer rows than features has it), and the oracle's
    posterior of it (`cols` None: the whole inverse; else those columns of it and the mean) -- made once per size."""
    if F not in _CASES:
        _CASES.clear()  # one size at a time in memory: 2 GiB per matrix at F = 16384
        rs = np.random.RandomState(F)
        B = rs.standard_normal((max(F // 8, 8), F))
        G = B.T @ B
        b = B.T @ rs.standard_normal(B.shape[0])
        iL = 1.0 / rs.gamma(2.0, 1.0, F)
        var = 0.37
        iC = np.diag(iL) + G / var
        if F <= F_FULL_INVERSE:
            cols = None
            Ch, ld_iC = orc.solve_posdef(iC, np.eye(F))
            mh = Ch @ b / var
        else:
            cols = np.sort(rs.choice(F, NCOLS, replace=False))
            rhs = np.zeros((F, NCOLS + 1))
            rhs[cols, np.arange(NCOLS)] = 1.0
            rhs[:, NCOLS] = b / var
            sol, ld_iC = orc.solve_posdef(iC, rhs)
            Ch, mh = sol[:, :NCOLS], sol[:, NCOLS]
        del iC
        _CASES[F] = (G, b, iL, var, Ch, mh, ld_iC, cols)
    return _CASES[F]


def _posterior(dev, _hip, F, G, b, iL, var):
    acc = dev.upload_vector(np.concatenate((G.ravel(), b)))
    dC = dev.malloc(F * F * 8)
    try:
        pG, pb = _hip.ctypes.c_void_p(acc.ptr.value), _hip.ctypes.c_void_p(acc.ptr.value + F * F * 8)
        post = dev.posterior(F, pG, pb, iL, var, dC)
        C = dev.download(dC, (F, F), np.float64) if post is not None else None
    finally:
        acc.free()
        dC.free()
    return post, C


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("mode", ["default", "deterministic"])
@pytest.mark.parametrize("F", [4096, 8257, F_LARGE])
def test_posterior_vs_oracle_solve_posdef(F, mode):
    from revrand_amd import _hip
    dev = _hip.get_device()
    G, b, iL, var, Ch, mh, ld_iC, cols = _case(F)
    was = dev.deterministic
    dev.set_deterministic(mode == "deterministic")
    try:
        (m, dg, logdet, tr), C = _posterior(dev, _hip, F, G, b, iL, var)
        if mode == "deterministic":  # the same bits on a second call
            (m2, dg2, logdet2, tr2), C2 = _posterior(dev, _hip, F, G, b, iL, var)
            assert np.array_equal(C, C2) and np.array_equal(m, m2) and logdet == logdet2 and tr == tr2
            del C2
    finally:
        dev.set_deterministic(was)
    assert np.array_equal(C, C.T)
    if cols is None:
        assert normwise(C, Ch) < 1e-9 and normwise(dg, Ch.diagonal()) < 1e-9
        trh = float((G * Ch).sum())
    else:   # the oracle's columns of the inverse (C is symmetric: rows too), the diagonal where they cross it
        assert normwise(C[:, cols], Ch) < 1e-9 and normwise(dg[cols], Ch[cols, np.arange(len(cols))]) < 1e-9
        assert np.array_equal(dg, C.diagonal())
        trh = float(np.einsum("ij,ij->", G, C))
    assert normwise(m, mh) < 1e-9
    assert abs(logdet - ld_iC) < 1e-9 * abs(ld_iC)
    assert abs(tr - trh) < 1e-9 * abs(trh)


def _breaks_in_panel(F, k, seed):
    """iC = U^T U with a well-conditioned random upper factor, except that pivot k is 1e-6 (below CHOLTHRESH = 1e-5): the
    first k pivots are fine, everything the panel pipeline does before column k is ordinary work."""
    rs = np.random.RandomState(seed)
    U = np.triu(rs.standard_normal((F, F)) / np.sqrt(F))
    U[np.arange(F), np.arange(F)] = 1.0 + rs.random_sample(F)
    U[k, k] = 1e-6
    return U.T @ U


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["default", "deterministic"])
@pytest.mark.parametrize("F,k", [(4096, 17 * 128 + 5), (4096, 31 * 128 + 127), (8257, 64 * 128 + 64)])
def test_not_positive_definite_in_a_late_panel_is_reported_and_leaves_nothing_in_flight(F, k, mode):
    """First bad pivot in panel >= 17 (of 32), in the very last column, and in the ragged last panel of F = 8257: `None`
    (RR_ERR_NOT_POSDEF) -- as the oracle's Cholesky path refuses the same matrix -- and the next call on a good matrix,
    issued straight away on the same context, is right: nothing of the failed call was left running on the second or third
    stream over the shared work space."""
    from revrand_amd import _hip
    dev = _hip.get_device()
    iC = _breaks_in_panel(F, k, F + k)
    # the reference's decision on this matrix: the Cholesky route is refused (pivot below CHOLTHRESH, or not factorisable)
    import scipy.linalg as sla
    try:
        refused = bool(np.any(sla.cholesky(iC, lower=False).diagonal() < orc.CHOLTHRESH))
    except sla.LinAlgError:
        refused = True
    assert refused
    iL, var = np.full(F, 0.5), 1.0
    G = iC - np.diag(iL)
    b = np.ones(F)
    was = dev.deterministic
    dev.set_deterministic(mode == "deterministic")
    try:
        post, _ = _posterior(dev, _hip, F, G, b, iL, var)
        assert post is None
        assert b"not safely positive definite" in dev.lib.rr_last_error()
        # the good matrix of the same size right behind it
        Gg, bg, iLg, varg, Ch, mh, ld_iC, cols = _case(F)
        (m, dg, logdet, tr), C = _posterior(dev, _hip, F, Gg, bg, iLg, varg)
        dev.sync()
    finally:
        dev.set_deterministic(was)
    assert normwise(C if cols is None else C[:, cols], Ch) < 1e-9 and normwise(m, mh) < 1e-9 and abs(logdet - ld_iC) < 1e-9 * abs(ld_iC)


@pytest.mark.parametrize("mode", ["default", "deterministic"])
@pytest.mark.parametrize("F", [1, 5, 64, 100, 256, 300, 512, 1000, 1024])
def test_small_posterior_in_one_launch_vs_oracle(F, mode, monkeypatch):
    """(opt-in route, RR_POSDEF_SMALL=1: measured slower than the panel pipeline at F = 512 and kept for its next attempt.)
    F <= 1024 (BASELINE config 1 is F = 512): rr_posterior_coop_kernel -- factor, inverse and C in ONE cooperative launch
    (32 workgroups, device-scope barriers) -- against the oracle's solve_posdef (mathfun/linalg.py:84-125): C, m, diag C,
    log|iC|, sum(G o C); ragged F (padding to 64-column panels), F below one panel."""
    from revrand_amd import _hip
    monkeypatch.setenv("RR_POSDEF_SMALL", "1")
    dev = _hip.get_device()
    rs = np.random.RandomState(F)
    B = rs.standard_normal((max(F // 2, 3), F))
    G = B.T @ B
    b = B.T @ rs.standard_normal(B.shape[0])
    iL = 1.0 / rs.gamma(2.0, 1.0, F)
    var = 0.37
    Ch, ld_iC = orc.solve_posdef(np.diag(iL) + G / var, np.eye(F))
    was = dev.deterministic
    dev.set_deterministic(mode == "deterministic")
    try:
        (m, dg, logdet, tr), C = _posterior(dev, _hip, F, G, b, iL, var)
        (m2, dg2, logdet2, tr2), C2 = _posterior(dev, _hip, F, G, b, iL, var)
    finally:
        dev.set_deterministic(was)
    assert np.array_equal(C, C2) and np.array_equal(m, m2) and logdet == logdet2   # fixed-order sums: the same bits
    assert np.array_equal(C, C.T)
    assert normwise(C, Ch) < 1e-10 and normwise(m, Ch @ b / var) < 1e-10 and normwise(dg, Ch.diagonal()) < 1e-10
    assert abs(logdet - ld_iC) < 1e-10 * max(abs(ld_iC), 1.0)
    trh = float((G * Ch).sum())
    assert abs(tr - trh) < 1e-9 * abs(trh)


@pytest.mark.parametrize("F,k", [(512, 500), (300, 10), (1024, 700)])
def test_small_posterior_reports_a_matrix_that_is_not_positive_definite(F, k, monkeypatch):
    """A pivot below CHOLTHRESH (or negative) in a late panel: RR_ERR_NOT_POSDEF (`None`), nothing left in flight, and the
    next call on a good matrix is right (the cooperative kernel's workgroups all leave behind the factorisation)."""
    from revrand_amd import _hip
    monkeypatch.setenv("RR_POSDEF_SMALL", "1")
    dev = _hip.get_device()
    iC = _breaks_in_panel(F, k, F + k)
    iL, var = np.full(F, 0.5), 1.0
    post, _ = _posterior(dev, _hip, F, iC - np.diag(iL), np.ones(F), iL, var)
    assert post is None and b"not safely positive definite" in dev.lib.rr_last_error()
    iCn = iC.copy()
    iCn[k, k] = -1.0 + iC[k, k]     # an indefinite matrix: a negative pivot
    post, _ = _posterior(dev, _hip, F, iCn - np.diag(iL), np.ones(F), iL, var)
    assert post is None
    rs = np.random.RandomState(1)
    B = rs.standard_normal((F // 2, F))
    G = B.T @ B
    (m, dg, logdet, tr), C = _posterior(dev, _hip, F, G, np.ones(F), iL, var)
    Ch, ld = orc.solve_posdef(np.diag(iL) + G / var, np.eye(F))
    assert normwise(C, Ch) < 1e-10 and abs(logdet - ld) < 1e-10 * abs(ld)
