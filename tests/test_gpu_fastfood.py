"""GPU parity of FastFoodRBF (transform / _makeVX / grad / gram) and `hadamard` against the
reference's golden vectors and the NumPy oracle."""
import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise

pytestmark = pytest.mark.gpu
TOL = {"f32": 1e-3, "f64": 1e-5}


def _ff(d, nb, ard, dtype, seed=3):
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    if ard:
        return bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=seed, dtype=dtype,
                              lenscale=Parameter(np.ones(d), Positive()))
    return bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=seed, dtype=dtype)


def test_hadamard_golden(golden):
    from revrand_amd.linalg import hadamard
    g = golden("hadamard")
    # the reference's own doctest vector (mathfun/linalg.py:202-206) -- exact in binary
    assert np.array_equal(hadamard(g["doctest_in"], ordering=False), g["doctest_nat"])
    assert np.array_equal(hadamard(g["doctest_in"], ordering=True), g["doctest_seq"])
    assert normwise(hadamard(g["Y"], ordering=False), g["nat"]) < 1e-14
    assert normwise(hadamard(g["Y"], ordering=True), g["seq"]) < 1e-14
    for L in (1, 2, 64, 128):
        assert normwise(hadamard(g["Y%d" % L], ordering=False), g["nat%d" % L]) < 1e-14
    Y = np.random.RandomState(0).randn(300, 1024).astype(np.float32)
    assert normwise(hadamard(Y, ordering=False), orc.hadamard(Y, False)) < 1e-5
    # involution: H(H(y)) = y / n
    assert normwise(hadamard(hadamard(Y, False), False) * 1024, Y) < 1e-5
    with pytest.raises(AssertionError):
        hadamard(np.zeros((2, 12)))


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("case", [(1, 10), (2, 10), (5, 16), (16, 64), (128, 256)])
def test_fastfood_golden(golden, case, dtype):
    d, nb = case
    g = golden("fastfood")
    k = "d%d_nb%d" % (d, nb)
    b = _ff(d, nb, False, dtype)
    # host sampling reproduces the reference's B, G, PI, S bit-for-bit (draw order B->G->PI->S)
    assert np.array_equal(b.B, g[k + "_B"]) and np.array_equal(b.PI, g[k + "_PI"])
    assert np.array_equal(b.G, g[k + "_G"]) and normwise(b.S, g[k + "_S"]) < 1e-14
    X = g[k + "_X"]
    assert normwise(b._makeVX(X), g[k + "_VX"]) < TOL[dtype] * 1e-2
    for tag, ls in [("iso0.7", 0.7), ("iso2.0", 2.0)]:
        P, dP = b.transform(X, ls), b.grad(X, ls)
        assert P.dtype == np.float64 and P.shape == g["%s_%s_Phi" % (k, tag)].shape
        assert normwise(P, g["%s_%s_Phi" % (k, tag)]) < TOL[dtype]
        assert normwise(dP, g["%s_%s_dPhi" % (k, tag)]) < TOL[dtype]
    if k + "_ard_Phi" in g:
        ba = _ff(d, nb, True, dtype)
        ls = np.linspace(0.5, 2.0, d)
        assert normwise(ba.transform(X, ls), g[k + "_ard_Phi"]) < TOL[dtype]
        dP = ba.grad(X, ls)
        assert dP.shape == g[k + "_ard_dPhi"].shape
        assert normwise(dP, g[k + "_ard_dPhi"]) < TOL[dtype]


@pytest.mark.parametrize("shape", [(1000, 3, 7), (513, 20, 100), (300, 64, 200), (257, 100, 256), (200, 128, 1024)])
def test_fastfood_vs_oracle(shape):
    N, d, nb = shape
    rs = np.random.RandomState(N)
    X = rs.randn(N, d).astype(np.float32)
    b = _ff(d, nb, True, "f32", seed=11)
    ls = np.linspace(0.6, 1.7, d)
    B, G, PI, S = orc.fastfood_matrices(nb, d, 11)
    ref = orc.fastfood_transform(X.astype(np.float64), B, G, PI, S, ls)
    assert normwise(b.transform(X, ls), ref) < 1e-3
    y = rs.randn(N).astype(np.float32)
    Gm, bv, yty = b.gram(X, y, ls)
    Gr, br, tr = orc.gram_stats(ref, y.astype(np.float64))
    assert normwise(Gm, Gr) < 1e-3 and normwise(bv, br) < 1e-3 and abs(yty - tr) < 1e-5 * tr


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("shape", [(1, 17, 32), (2, 33, 64 * 3), (255, 63, 64 * 5), (130, 101, 128 * 7), (77, 129, 256 * 3),
                                   (64, 24, 32 * 6), (3, 16, 16 * 9)])
def test_fastfood_chain_kernel_ragged_paths(shape, dtype):
    """The lane-major chain kernels' other instances: input dimensions that are not a multiple of the vector width (scalar
    loads / stores, elements beyond d read from a clamped address), block counts that are not a multiple of 4 (part of
    a wave idle), odd row counts (the unrolled row loop repeats its last row), one and two rows."""
    N, d, nb = shape
    rs = np.random.RandomState(N + d)
    X = rs.randn(N, d)
    b = _ff(d, nb, True, dtype, seed=5)
    ls = np.linspace(0.6, 1.7, d)
    B, G, PI, S = orc.fastfood_matrices(nb, d, 5)
    ref = orc.fastfood_transform(X, B, G, PI, S, ls)
    P = b.transform(X, ls)
    assert P.shape == ref.shape and np.all(np.isfinite(P))
    assert normwise(P, ref) < TOL[dtype]
    dP = b.grad(X, ls)
    assert dP.shape == (N, ref.shape[1], d)
    if N <= 130 and d <= 101:
        assert normwise(dP, orc.fastfood_grad(X, B, G, PI, S, ls)) < TOL[dtype]


def test_fastfood_in_concat_and_slm():
    import revrand_amd.basis_functions as bs
    from revrand_amd.slm import StandardLinearModel
    rs = np.random.RandomState(4)
    X = rs.randn(300, 3)
    y = np.sin(X[:, 0]) + 0.05 * rs.randn(300)
    base = bs.FastFoodRBF(nbases=20, Xdim=3, random_state=1) + bs.LinearBasis(onescol=True)
    P = base.transform(X, 1.0)
    assert P.shape == (300, 2 * 20 + 4) and base.get_dim(X) == P.shape[1]
    # L-BFGS paths are sensitive to the last bits of the (atomically accumulated) statistics: judge the fit by what
    # is robust -- the ELBO did not get worse than at the initial parameters and the model beats the mean predictor
    slm = StandardLinearModel(base, nstarts=0, maxiter=100, random_state=2)  # seeded: the start point is an rvs draw
    slm.obj_ = -np.inf
    slm._elbo(X, y, 1.0, [1.0, 1.0], 1.0)
    elbo0 = slm.obj_
    slm.fit(X, y)
    assert slm.obj_ >= elbo0 - 1e-3 * abs(elbo0)   # (the two evaluations of the start point use different code paths)
    assert ((slm.predict(X) - y) ** 2).mean() < 0.7 * y.var()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("case", [(1, 10, 3), (3, 8, 4), (5, 16, 3), (16, 32, 3)])
def test_fastfood_gm_golden(golden, case, dtype):
    """FastFoodGM.transform / grad (mean and lenscale gradients) vs the reference's outputs."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Bound, Parameter, Positive
    d, nb, seed = case
    g = golden("fastfood_gm")
    k = "d%d_nb%d" % (d, nb)
    b = bs.FastFoodGM(nbases=nb, Xdim=d, random_state=seed, mean=Parameter(np.zeros(d), Bound()),
                      lenscale=Parameter(np.ones(d), Positive()), dtype=dtype)
    X, mean, ls = g[k + "_X"], g[k + "_mean"], g[k + "_ls"]
    P = b.transform(X, mean, ls)
    assert P.shape == g[k + "_Phi"].shape == (X.shape[0], 4 * b.n) and b.get_dim(X) == 4 * b.n
    assert normwise(P, g[k + "_Phi"]) < TOL[dtype]
    dM, dL = b.grad(X, mean, ls)
    assert dM.shape == g[k + "_dmean"].shape and dL.shape == g[k + "_dlen"].shape
    assert normwise(dM, g[k + "_dmean"]) < TOL[dtype] and normwise(dL, g[k + "_dlen"]) < TOL[dtype]
    # defaults (None) use the initial parameter values, scalars are broadcast to (d,)
    b2 = bs.FastFoodGM(nbases=nb, Xdim=d, random_state=seed, dtype=dtype)
    assert b2.params[0].shape == (d,) and b2.params[1].shape == (d,) and len(b2.params_values()) == 2
    assert b2.transform(X).shape == P.shape and bs.count_args(b2.transform) == 3


def test_all_bases_concatenate_like_reference_test_bases():
    """tests/test_bases.py:128-220 of the reference: every basis + one big concatenation, shapes only."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Bound, Parameter, Positive
    from functools import reduce
    from operator import add
    rs = np.random.RandomState(0)
    N, d = 50, 2
    X = rs.randn(N, d)
    ard = Parameter(np.ones(d), Positive())
    bases = [bs.BiasBasis(), bs.LinearBasis(onescol=True), bs.RandomRBF(Xdim=d, nbases=10),
             bs.RandomRBF(Xdim=d, nbases=10, lenscale=ard), bs.OrthogonalRBF(Xdim=d, nbases=10),
             bs.OrthogonalRBF(Xdim=d, nbases=10, lenscale=ard), bs.FastFoodRBF(Xdim=d, nbases=10),
             bs.FastFoodRBF(Xdim=d, nbases=10, lenscale=ard), bs.FastFoodGM(Xdim=d, nbases=10),
             bs.FastFoodGM(Xdim=d, nbases=10, mean=Parameter(np.zeros(d), Bound()), lenscale=ard)]
    hypers = [(), (), (1.,), (np.ones(d),), (1.,), (np.ones(d),), (1.,), (np.ones(d),),
              (np.ones(d), np.ones(d)), (np.ones(d), np.ones(d))]
    for b, h in zip(bases, hypers):
        P = b.transform(X, *h)
        assert P.shape[0] == N and P.ndim == 2
    bcat = reduce(add, bases)
    hyps = [v for h in hypers for v in h]
    P = bcat.transform(X, *hyps)
    assert bcat.get_dim(X) == P.shape[1]
    grads = list(bcat.grad(X, *hyps))
    assert len(grads) == 10 and all(g.shape[:2] == P.shape for g in grads)   # 6 lenscales + 2x(mean, lenscale)


def test_config4_full_width_properties():
    """BASELINE config 4's width (FastFoodRBF nbases=8192, D=128 -> F=16384), 20k rows, through size-independent
    properties: unit row norms (cos^2 + sin^2), linearity of the structured projection, the FWHT chain against its
    dense equivalent through the MFMA feature kernel, and a few rows against the oracle's NumPy chain."""
    import revrand_amd.basis_functions as bs
    N, d, nb = 20_000, 128, 8192
    rs = np.random.RandomState(4)
    X = rs.randn(N, d).astype(np.float32)
    b = bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=6)
    Phi = b.transform(X, 1.3)
    assert Phi.shape == (N, 2 * nb) and Phi.dtype == np.float64
    assert np.abs((Phi ** 2).sum(axis=1) - 1.0).max() < 1e-5
    ff, rff = b._handles()
    V1, V2 = ff.vx(X[:512], 1.0), ff.vx(X[512:1024], 1.0)
    V12 = ff.vx(X[:512] + 2 * X[512:1024], 1.0)
    assert normwise(V12, V1 + 2 * V2) < 1e-5
    dense = rff.transform(X[:2048], 1.3)                     # W = _makeVX(I) through rr_rff_features_mfma_kernel
    assert normwise(dense, Phi[:2048]) < 1e-4
    B, G, PI, S = orc.fastfood_matrices(nb, d, 6)
    assert normwise(Phi[:64], orc.fastfood_transform(X[:64].astype(np.float64), B, G, PI, S, 1.3)) < 1e-3


def test_fastfood_transform_device_resident():
    """rr_fastfood_transform_dev: X and Phi stay in HBM; same values as the host-buffer call, f32 and f64 output,
    ragged row count and an output leading dimension wider than 2n."""
    import revrand_amd.basis_functions as bs
    rs = np.random.RandomState(1)
    X = rs.randn(777, 21)
    b = bs.FastFoodRBF(nbases=40, Xdim=21, random_state=2)
    ff, _ = b._handles()
    dev = ff.dev
    F = 2 * ff.n
    want = ff.transform(X, 0.8)
    dX = dev.upload_matrix(X.astype(np.float32))
    for dt, ld in ((np.float32, F), (np.float64, F + 5)):
        out = dev.zeros(777 * ld * np.dtype(dt).itemsize)
        ff.transform_dev(dX, 0.8, out, out_dtype=dt, ldphi=ld)
        got = dev.download(out, (777, ld), dt)
        assert normwise(got[:, :F], want) < 1e-5 and np.all(got[:, F:] == 0)
        out.free()


def test_fastfood_wide_input():
    """128 < d <= 256: the FWHT chain kernel serves `transform` / `_makeVX` (d2 = 256); grad and the Gram go through the
    dense-equivalent random Fourier handle (GEMM route for Xdim > 128, tests/test_gpu_large_xdim.py)."""
    import revrand_amd.basis_functions as bs
    rs = np.random.RandomState(0)
    X = rs.randn(300, 200)
    y = rs.randn(300)
    f = bs.FastFoodRBF(nbases=300, Xdim=200, random_state=1)
    B, G, PI, S = orc.fastfood_matrices(300, 200, 1)
    want = orc.fastfood_transform(X, B, G, PI, S, 1.7)
    assert normwise(f.transform(X, 1.7), want) < 1e-3
    Gm, bv, _ = f.gram(X, y, 1.7)
    assert normwise(Gm, want.T @ want) < 1e-3 and normwise(bv, want.T @ y) < 1e-3


def _oracle_ff_elbo(f, X, y, var, reg, ls, extra=None):
    """slm_elbo of the oracle on FastFood features (+ an optional block of parameter-free columns)."""
    X64, y64 = X.astype(np.float64), y.astype(np.float64)
    Phi = orc.fastfood_transform(X64, f.B, f.G, f.PI, f.S, ls)
    dP = orc.fastfood_grad(X64, f.B, f.G, f.PI, f.S, ls)
    slabs = [dP] if dP.ndim == 2 else [dP[:, :, i] for i in range(dP.shape[2])]
    regs = np.atleast_1d(np.asarray(reg, dtype=float))
    if extra is None:
        return orc.slm_elbo(Phi, y64, var, np.full(Phi.shape[1], regs[0]), slice(None), slabs)
    full = np.hstack((Phi, extra))
    pad = [np.hstack((s, np.zeros_like(extra))) for s in slabs]
    diag = np.concatenate((np.full(Phi.shape[1], regs[0]), np.full(extra.shape[1], regs[1])))
    return orc.slm_elbo(full, y64, var, diag, [slice(0, Phi.shape[1]), slice(Phi.shape[1], full.shape[1])], pad)


@pytest.mark.parametrize("shape", [(700, 5, 16, True), (515, 3, 40, False), (1300, 20, 256, True), (900, 128, 512, True),
                                   (600, 100, 300, True)])
def test_fastfood_resident_elbo_runs_the_chain_and_matches_the_oracle(shape, monkeypatch):
    """StandardLinearModel._elbo on a FastFoodRBF basis with (X, y) resident (basis_functions.py:1263-1289 feeding
    slm.py:142-199): the statistics pass is the chain kernel writing Phi into the device feature matrix + the MFMA SYRK,
    the second pass contracts X^T A against the same features (n % 256 == 0: fused in registers; otherwise the stored
    route), the dense equivalent W only meets T on the host.  Against the oracle's FWHT chain in float64, and against the
    dense-equivalent route (RR_FASTFOOD_FIT=dense) the fit used before."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.slm import StandardLinearModel
    N, d, nb, ard = shape
    rs = np.random.RandomState(N + d)
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X @ rs.randn(d) / np.sqrt(d)) + 0.1 * rs.randn(N)).astype(np.float32)
    f = _ff(d, nb, ard, "f32")
    ls = np.linspace(0.8, 1.4, d) if ard else 1.2
    var, reg = 0.3, 1.5
    out = {}
    for route in ("chain", "dense"):
        monkeypatch.setenv("RR_FASTFOOD_FIT", route)
        slm = StandardLinearModel(f)
        slm.obj_ = -np.inf
        slm._state = slm._make_state(X, y)
        assert type(slm._state).__name__ == ("CatFitState" if route == "chain" else "DeviceFitState")
        try:
            nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, var, reg, ls)
            C = slm._state.best_covariance() if getattr(slm._state, "best_on_device", False) else slm.covariance_
        finally:
            slm._state.release()
            slm._state = None
        out[route] = (nelbo, ndvar, ndreg, np.atleast_1d(ndhyp), slm.weights_.copy(), np.array(C))
    ref = _oracle_ff_elbo(f, X, y, var, reg, ls)
    for route, (nelbo, ndvar, ndreg, ndhyp, m, C) in out.items():
        assert abs(-nelbo - ref["elbo"]) < 1e-4 * abs(ref["elbo"]), route
        assert abs(-ndvar - ref["dvar"]) < 1e-3 * abs(ref["dvar"]) and abs(-ndreg - ref["dreg"][0]) < 1e-3 * abs(ref["dreg"][0])
        assert normwise(-ndhyp, np.array(ref["dhyp"])) < 2e-3, route
        assert normwise(m, ref["m"]) < 1e-3 and normwise(C, ref["C"]) < 1e-3
    assert normwise(out["chain"][3], out["dense"][3]) < 2e-3


def test_fastfood_child_of_a_concatenation_is_resident():
    """BasisCat(FastFoodRBF + LinearBasis): the FastFood child takes part in the device-resident fit (its block of the
    feature matrix from the chain kernel); `_elbo` against the oracle on the hstacked features."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    rs = np.random.RandomState(9)
    N, d, nb = 800, 6, 256
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X[:, 0]) + 0.3 * X[:, 1] + 0.1 * rs.randn(N)).astype(np.float32)
    f = bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()))
    cat = f + bs.LinearBasis(onescol=True)
    slm = StandardLinearModel(cat)
    slm.obj_ = -np.inf
    slm._state = slm._make_state(X, y)
    assert type(slm._state).__name__ == "CatFitState" and type(slm._state.children[0]).__name__ == "_ResidentFastFood"
    ls = np.linspace(0.9, 1.3, d)
    try:
        nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, 0.4, [1.2, 0.7], ls)
    finally:
        slm._state.release()
        slm._state = None
    ref = _oracle_ff_elbo(f, X, y, 0.4, [1.2, 0.7], ls, extra=orc.linear_transform(X.astype(np.float64), True))
    assert abs(-nelbo - ref["elbo"]) < 1e-4 * abs(ref["elbo"])
    assert normwise(-np.asarray(ndreg), np.array(ref["dreg"])) < 1e-3 and normwise(-np.asarray(ndhyp), np.array(ref["dhyp"])) < 2e-3
    assert normwise(slm.weights_, ref["m"]) < 1e-3


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_fastfood_extreme_length_scales_give_finite_features_like_the_reference(dtype):
    """The optimiser's log-space bounds let L-BFGS-B try length scales of 1e-100 and 1e+100 (optimize/decorators.py:18):
    the reference's float64 chain returns finite noise / the constant features there, and so must the kernels -- a NaN in
    Phi becomes a NaN objective and scipy's Cholesky raises instead of the line search backing off."""
    f = _ff(3, 20, False, dtype)
    X = np.random.RandomState(4).randn(300, 3)
    for ls in (9.99999999999989e-101, 1e100):
        P = f.transform(X, ls)
        ref = orc.fastfood_transform(X, f.B, f.G, f.PI, f.S, ls)
        assert np.isfinite(ref).all() and np.isfinite(P).all() and np.abs(P).max() <= 1.0 / np.sqrt(f.n) * (1 + 1e-6)
        assert np.abs((P ** 2).sum(axis=1) - 1.0).max() < 1e-5   # cos^2 + sin^2 over n frequencies, / n
    assert normwise(f.transform(X, 1e100), orc.fastfood_transform(X, f.B, f.G, f.PI, f.S, 1e100)) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# FastFoodGM as a first-class basis (reference: basis_functions.py:1386-1562)
# ---------------------------------------------------------------------------------------------------------------------

def _gm(d, nb, dtype="f32", seed=5):
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Bound, Parameter, Positive
    return bs.FastFoodGM(nbases=nb, Xdim=d, random_state=seed, mean=Parameter(np.zeros(d), Bound()),
                         lenscale=Parameter(np.ones(d), Positive()), dtype=dtype)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("d,nb,N", [(16, 64, 1001), (100, 1024, 777), (128, 8192, 131), (9, 40, 64)])
def test_fastfood_gm_chain_vs_oracle_at_scale(d, nb, N, dtype):
    """FastFoodGM.transform through the chain kernel's mixture mode (transform :1443-1475) against the oracle's FWHT chain
    in float64 at the sizes the configurations use -- d2 = 16 / 128 (ragged d = 100: x padded in registers) / 128 with
    k = 64 blocks, odd row counts -- with mean shifts of several revolutions (|x . mean| ~ 2 pi x 5 at d = 128: the
    float32 phase of VX +- mX is reduced modulo one revolution AFTER the shift is added), and against the dense-equivalent
    route (rr_gm_transform) it replaces."""
    rs = np.random.RandomState(d + nb)
    b = _gm(d, nb, dtype)
    X = rs.randn(N, d).astype(np.float32 if dtype == "f32" else np.float64)
    mean, ls = 0.8 * rs.randn(d), np.linspace(0.7, 1.6, d)
    ff, dense = b._handles()
    assert ff.gm_chain_ok
    P = b.transform(X, mean, ls)
    ref = orc.fastfood_gm_transform(X.astype(np.float64), b.B, b.G, b.PI, b.S, mean, ls)
    assert P.shape == ref.shape == (N, 4 * b.n)
    assert np.abs(X.astype(np.float64) @ mean).max() > 6.0        # the shift is worth more than a revolution
    assert normwise(P, ref) < TOL[dtype]
    assert normwise(dense.gm_transform(X, mean, ls), ref) < TOL[dtype] * (10 if dtype == "f32" else 1)
    # zero mean: the two halves coincide and equal FastFoodRBF's features / sqrt(2)
    P0 = b.transform(X, np.zeros(d), ls)
    n = b.n
    assert np.array_equal(P0[:, :2 * n], P0[:, 2 * n:])
    assert normwise(P0[:, :2 * n] * np.sqrt(2.0), orc.fastfood_transform(X.astype(np.float64), b.B, b.G, b.PI, b.S, ls)) < TOL[dtype]


def _oracle_gm_elbo(b, X, y, var, reg, mean, ls):
    X64 = X.astype(np.float64)
    Phi = orc.fastfood_gm_transform(X64, b.B, b.G, b.PI, b.S, mean, ls)
    dM, dL = orc.fastfood_gm_grad(X64, b.B, b.G, b.PI, b.S, mean, ls)
    slabs = [np.ascontiguousarray(dM[:, :, i]) for i in range(b.d)] + [np.ascontiguousarray(dL[:, :, i]) for i in range(b.d)]
    return orc.slm_elbo(Phi, y.astype(np.float64), var, np.full(Phi.shape[1], reg), slice(None), slabs)


@pytest.mark.parametrize("shape", [(64, 16, 256), (300, 20, 70), (1000, 32, 512)])
def test_fastfood_gm_resident_elbo_matches_the_oracle_chain(shape):
    """`_elbo` (slm.py:142-199) on a FastFoodGM basis with (X, y) resident: the statistics pass is the chain kernel's four
    blocks in the device feature matrix + the MFMA SYRK; the second pass contracts BOTH gradients (mean and lenscale,
    basis_functions.py:1477-1537) on the device as two random-Fourier shaped children (n % 256 == 0: in registers; else the
    stored route) -- no (N, 4n, d) tensor.  Against the oracle's chain + slm_elbo in float64."""
    from revrand_amd.slm import StandardLinearModel
    N, d, nb = shape
    rs = np.random.RandomState(N + d)
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X @ rs.randn(d) / np.sqrt(d)) + 0.1 * rs.randn(N)).astype(np.float32)
    b = _gm(d, nb)
    mean, ls = 0.3 * rs.randn(d), np.linspace(0.8, 1.4, d)
    var, reg = 0.3, 1.5
    slm = StandardLinearModel(b)
    slm.obj_ = -np.inf
    slm._state = slm._make_state(X, y)
    assert type(slm._state).__name__ == "CatFitState" and type(slm._state.children[0]).__name__ == "_ResidentFastFoodGM"
    try:
        nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, var, reg, [mean, ls])
        C = slm._state.best_covariance() if getattr(slm._state, "best_on_device", False) else slm.covariance_
    finally:
        slm._state.release()
        slm._state = None
    ref = _oracle_gm_elbo(b, X, y, var, reg, mean, ls)
    assert isinstance(ndhyp, list) and len(ndhyp) == 2 and ndhyp[0].shape == ndhyp[1].shape == (d,)
    assert abs(-nelbo - ref["elbo"]) < 1e-4 * abs(ref["elbo"])
    assert abs(-ndvar - ref["dvar"]) < 1e-3 * abs(ref["dvar"]) and abs(-ndreg - ref["dreg"][0]) < 1e-3 * abs(ref["dreg"][0])
    assert normwise(-ndhyp[0], np.array(ref["dhyp"][:d])) < 2e-3      # d ELBO / d mean
    assert normwise(-ndhyp[1], np.array(ref["dhyp"][d:])) < 2e-3      # d ELBO / d lenscale
    assert normwise(slm.weights_, ref["m"]) < 1e-3 and normwise(C, ref["C"]) < 1e-3
    # the fused statistics and the prediction route, against the same oracle features
    G, bv, yty = b.gram(X, y, mean, ls)
    Phi = orc.fastfood_gm_transform(X.astype(np.float64), b.B, b.G, b.PI, b.S, mean, ls)
    assert normwise(G, Phi.T @ Phi) < 1e-3 and normwise(bv, Phi.T @ y.astype(np.float64)) < 1e-3
    Ey, Vf = b.predict_moments(X[:50], [mean, ls], ref["m"], ref["C"])
    Eo, Vo = orc.slm_predict_moments(Phi[:50], ref["m"], ref["C"], 0.0)
    assert normwise(Ey, Eo) < 1e-3 and normwise(Vf, Vo) < 1e-3


def test_fastfood_gm_fits_alone_and_in_a_spectral_mixture():
    """A spectral mixture as the reference builds it (basis_functions.py:1394-1396: "concatenate as many of these objects
    as desired"): two FastFoodGM components + a LinearBasis fit device-resident (every child takes part), gradients of the
    concatenation in the reference's order [mean_1, ls_1, mean_2, ls_2]; a lone component fits too; sklearn can clone it."""
    import revrand_amd.basis_functions as bs
    from sklearn.base import clone
    from revrand_amd.slm import StandardLinearModel
    rs = np.random.RandomState(4)
    N, d = 4000, 16
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(2.0 * X[:, 0]) * np.cos(X[:, 1]) + 0.1 * rs.randn(N)).astype(np.float32)
    cat = _gm(d, 64, seed=1) + _gm(d, 64, seed=2) + bs.LinearBasis(onescol=True)
    slm = StandardLinearModel(cat, nstarts=0, maxiter=12, random_state=0)
    slm.obj_ = -np.inf
    slm._state = slm._make_state(X, y)
    assert type(slm._state).__name__ == "CatFitState"
    hyp = [0.2 * rs.randn(d), np.ones(d), -0.2 * rs.randn(d), 1.3 * np.ones(d)]
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, 0.5, [1.0, 1.2, 0.7], hyp)
    slm._state.release()
    slm._state = None
    assert len(ndhyp) == 4 and all(g.shape == (d,) for g in ndhyp)
    # against the oracle on 4n + 4n + (d + 1) hstacked features
    X64 = X.astype(np.float64)
    blocks, slabs = [], []
    for gm, (mu, ls) in zip(cat.bases[:2], (hyp[:2], hyp[2:])):
        blocks.append(orc.fastfood_gm_transform(X64, gm.B, gm.G, gm.PI, gm.S, mu, ls))
    blocks.append(orc.linear_transform(X64, True))
    Phi = np.hstack(blocks)
    ends = np.cumsum([0] + [bl.shape[1] for bl in blocks])
    for i, (gm, (mu, ls)) in enumerate(zip(cat.bases[:2], (hyp[:2], hyp[2:]))):
        for dP in orc.fastfood_gm_grad(X64, gm.B, gm.G, gm.PI, gm.S, mu, ls):
            for k in range(d):
                full = np.zeros_like(Phi)
                full[:, ends[i]:ends[i + 1]] = dP[:, :, k]
                slabs.append(full)
    diag = np.concatenate([np.full(ends[i + 1] - ends[i], r) for i, r in enumerate([1.0, 1.2, 0.7])])
    ref = orc.slm_elbo(Phi, y.astype(np.float64), 0.5, diag, [slice(int(ends[i]), int(ends[i + 1])) for i in range(3)], slabs)
    assert abs(-nelbo - ref["elbo"]) < 1e-4 * abs(ref["elbo"])
    assert normwise(-np.concatenate(ndhyp), np.array(ref["dhyp"])) < 2e-3
    assert normwise(-np.array(ndreg), np.array(ref["dreg"])) < 1e-3
    # end to end
    fitted = clone(slm).fit(X, y)
    assert np.isfinite(fitted.obj_) and ((fitted.predict(X) - y) ** 2).mean() < 0.8 * y.var()
    Ey, Vy = fitted.predict_moments(X[:100])
    assert np.all(np.isfinite(Ey)) and np.all(Vy > 0)
    lone = StandardLinearModel(_gm(d, 128, seed=3), nstarts=0, maxiter=10, random_state=0).fit(X, y)
    assert np.isfinite(lone.obj_) and len(lone.hypers_) == 2
