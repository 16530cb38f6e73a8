"""
GPU parity of the random Fourier path for input dimension Xdim > 128 (the reference has no limit; its own
classification demo runs on 256-pixel USPS digits): phases through a GEMM (f32) or the f64 phase kernel, then the
trig kernel -- against the NumPy oracle, for every consumer (transform, grad, Gram, second `_elbo` pass,
predict_moments, GLM minibatch step, gradient contraction, FastFood's dense equivalent).
"""
import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise

pytestmark = pytest.mark.gpu

TOL = {"f32": 1e-3, "f64": 1e-5}


def _imports():
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    return bs, Parameter, Positive


def _data(N, d, seed):
    rs = np.random.RandomState(seed)
    X = rs.randn(N, d) / np.sqrt(d / 8.0)          # |x| ~ sqrt(8): phases of a few revolutions, like d = 8 data
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    return rs, X, y


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("shape", [(1, 129, 4), (700, 200, 130), (1031, 256, 300), (300, 1000, 64)])
def test_transform_and_grad_vs_oracle(shape, dtype):
    bs, Parameter, Positive = _imports()
    N, d, n = shape
    rs, X, _ = _data(N, d, N + d)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=5, dtype=dtype, lenscale=Parameter(np.ones(d), Positive()))
    ls = np.linspace(0.7, 1.9, d)
    P = b.transform(X, ls)
    assert P.shape == (N, 2 * n) and P.dtype == np.float64
    assert normwise(P, orc.rff_transform(X, b.W, ls)) < TOL[dtype]
    assert normwise(b.transform(X.astype(np.float32), ls), orc.rff_transform(X.astype(np.float32), b.W, ls)) < TOL["f32"]
    if N * n * d < 3e7:
        dP = b.grad(X, ls)
        assert dP.shape == (N, 2 * n, d)
        assert normwise(dP, orc.rff_grad(X, b.W, ls)) < TOL[dtype]
    bi = bs.RandomMatern32(nbases=n, Xdim=d, random_state=6, dtype=dtype)
    assert normwise(bi.transform(X, 1.4), orc.rff_transform(X, bi.W, 1.4)) < TOL[dtype]
    assert normwise(bi.grad(X, 1.4), orc.rff_grad(X, bi.W, 1.4)) < TOL[dtype]      # iso quirk: dimension 0 only


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("shape", [(1500, 200, 130), (5000, 256, 128), (257, 384, 40)])
def test_gram_vs_oracle(shape, dtype):
    bs, Parameter, Positive = _imports()
    N, d, n = shape
    rs, X, y = _data(N, d, N + n)
    X, y = X.astype(np.float32), y.astype(np.float32)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=2, dtype=dtype, lenscale=Parameter(np.ones(d), Positive()))
    ls = np.linspace(0.8, 1.6, d)
    G, bv, yty = b.gram(X, y, ls)
    Gr, br, ytyr = orc.rff_gram_chunked(X.astype(np.float64), y.astype(np.float64), b.W, ls)
    assert np.array_equal(G, G.T)
    assert normwise(G, Gr) < (2e-5 if dtype == "f32" else 1e-10)
    assert normwise(bv, br) < (1e-4 if dtype == "f32" else 1e-10)
    assert abs(yty - ytyr) < 1e-5 * ytyr


def test_many_row_subchunks_equal_one_pass():
    """More rows than one phase sub-chunk (131072): the seams must not show."""
    bs, Parameter, Positive = _imports()
    N, d, n = 140000, 130, 64
    rs, X, y = _data(N, d, 1)
    X, y = X.astype(np.float32), y.astype(np.float32)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=2)
    G, bv, _ = b.gram(X, y, 1.1)
    G1, b1, _ = b.gram(X[:131072], y[:131072], 1.1)
    G2, b2, _ = b.gram(X[131072:], y[131072:], 1.1)
    assert normwise(G, G1 + G2) < 1e-5 and normwise(bv, b1 + b2) < 1e-4
    P = b.transform(X[131000:131200], 1.1)
    assert normwise(P, orc.rff_transform(X[131000:131200], b.W, 1.1)) < 1e-3
    Pall = b.transform(X, 1.1)
    assert np.array_equal(Pall[131000:131200], P)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_second_pass_and_predict_vs_oracle(dtype):
    bs, Parameter, Positive = _imports()
    N, d, n = 900, 200, 70
    rs, X, y = _data(N, d, 4)
    basis = bs.RandomMatern52(nbases=n, Xdim=d, random_state=3, dtype=dtype, lenscale=Parameter(np.ones(d), Positive()))
    ls = np.linspace(0.7, 1.5, d)
    var, reg = 0.3, 1.4
    Phi = orc.rff_transform(X, basis.W, ls)
    dP = orc.rff_grad(X, basis.W, ls)
    o = orc.slm_elbo(Phi, y, var, np.full(2 * n, reg), slice(None), [dP[:, :, i] for i in range(d)])
    st = basis.device_fit_state(X, y)
    sq, dh = st.second_pass(ls, o["m"], o["C"], var)
    st.release()
    err = y - Phi @ o["m"]
    assert abs(sq - err @ err) < 1e-4 * (err @ err)
    assert normwise(dh, -np.array(o["dhyp"])) < 2e-3
    Xs = rs.randn(257, d) / np.sqrt(d / 8.0)
    Ey, Vf = basis.predict_moments(Xs, ls, o["m"], o["C"])
    Eo, Vo = orc.slm_predict_moments(orc.rff_transform(Xs, basis.W, ls), o["m"], o["C"], 0.0)
    assert normwise(Ey, Eo) < 1e-4 and normwise(Vf, Vo) < 1e-3


def test_slm_fit_on_256_dimensional_inputs():
    """End to end: resident fit (device posterior, F = 512) + predict on d = 256 inputs."""
    bs, Parameter, Positive = _imports()
    from revrand_amd import StandardLinearModel
    rs = np.random.RandomState(0)
    N, d = 3000, 256
    X = rs.randn(N + 500, d)
    w = np.zeros(d)
    w[:6] = rs.randn(6)
    f = np.sin(X @ w) + 0.3 * (X @ w)
    y = f + 0.1 * rs.randn(N + 500)
    basis = bs.RandomRBF(nbases=256, Xdim=d, random_state=1, lenscale=Parameter(16.0, Positive())) + bs.LinearBasis(onescol=True)
    slm = StandardLinearModel(basis, maxiter=60, random_state=0).fit(X[:N], y[:N])
    Ey = slm.predict(X[N:])
    smse = ((y[N:] - Ey) ** 2).mean() / y[N:].var()
    assert np.isfinite(Ey).all() and smse < 0.5, smse


@pytest.mark.parametrize("ard", [False, True])
def test_grad_contract_vs_materialised_gradient(ard):
    bs, Parameter, Positive = _imports()
    N, d, n = 1200, 150, 90
    rs, X, _ = _data(N, d, 2)
    E = rs.randn(N, 2 * n)
    b = bs.RandomCauchy(nbases=n, Xdim=d, random_state=4, lenscale=Parameter(np.ones(d) if ard else 1., Positive()))
    ls = np.linspace(0.7, 1.6, d) if ard else 1.3
    got = b.grad_contract(X, E, ls)
    dP = orc.rff_grad(X, b.W, ls)
    want = np.array([(E * dP[:, :, i]).sum() for i in range(d)]) if ard else (E * dP).sum()
    assert np.shape(got) == np.shape(want)
    assert normwise(np.atleast_1d(got), np.atleast_1d(want)) < 1e-3


def test_glm_minibatch_step_vs_oracle():
    """The SVI step on a d = 200 ARD basis + bias column: objective, dm, dC and the 200 length-scale gradients."""
    bs, Parameter, Positive = _imports()
    import revrand_amd.likelihoods as lk
    from revrand_amd import GeneralizedLinearModel as GLM
    rs = np.random.RandomState(3)
    M, d, K, L = 333, 200, 3, 5
    X = rs.randn(M, d) / np.sqrt(d / 8.0)
    y = rs.poisson(np.exp(0.5 * np.sin(X[:, 0] * 3))).astype(float)
    cat = bs.RandomRBF(nbases=40, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True)
    ls0 = np.linspace(0.8, 1.3, d)
    hyp = [ls0]
    Phi = cat.transform(X, *hyp)
    dPs = []
    for gfull in cat.grad(X, *hyp):
        dPs.extend([gfull[:, :, i] for i in range(gfull.shape[2])] if np.ndim(gfull) == 3 else ([gfull] if np.size(gfull) else []))
    D = Phi.shape[1]
    m = 0.2 * rs.randn(D, K)
    C = rs.gamma(2., 0.5, size=(D, K))
    regs = [1.2, 0.8]
    Ld, slices = cat.regularizer_diagonal(X, *regs)
    e = np.stack([np.random.RandomState(9).randn(K * L, D)[k * L:(k + 1) * L] for k in range(K)])
    want = orc.glm_elbo(m, C, Ld, slices, "poisson_exp", [], (), Phi, dPs, y, e, 6.0)
    glm = GLM(likelihood=lk.Poisson(), basis=cat, K=K, nsamples=L, random_state=9)
    glm.B_, glm.D_ = 6.0, D
    glm._GeneralizedLinearModel__it = -1
    nobj, (ndm, ndC, dL, dlp, dbp) = glm._elbo(m.copy(), C.copy(), regs, [], hyp, X, y)
    glm._release_features()
    assert abs(nobj - want[0]) < 2e-4 * abs(want[0])
    assert normwise(ndm, want[1][0]) < 1e-3 and normwise(ndC, want[1][1]) < 1e-3
    flat = np.concatenate([np.atleast_1d(v) for v in dbp])
    assert flat.shape == (d,)
    assert normwise(flat, np.array(want[1][4])) < 2e-3


def test_fastfood_dense_equivalent_for_wide_inputs():
    """FastFoodRBF with d = 200 (d2 = 256): grad and Gram run on the dense equivalent of the chain."""
    bs, Parameter, Positive = _imports()
    N, d = 400, 200
    rs, X, y = _data(N, d, 7)
    b = bs.FastFoodRBF(nbases=300, Xdim=d, random_state=3)
    P = b.transform(X, 1.3)
    G, bv, _ = b.gram(X, y, 1.3)
    assert normwise(G, P.T @ P) < 1e-4 and normwise(bv, P.T @ y) < 1e-4
    dP = b.grad(X, 1.3)
    V = b._makeVX(np.eye(d))
    assert normwise(dP, orc.rff_grad(X, V, 1.3)) < 1e-3


def test_xdim_limit_is_reported():
    bs, _, _ = _imports()
    b = bs.RandomRBF(nbases=8, Xdim=5000, random_state=0)
    with pytest.raises((ValueError, RuntimeError), match="Xdim=5000"):
        b.transform(np.zeros((4, 5000)))
