"""GLM SVI step on the MI355X against the reference's golden minibatch `_elbo`, the
oracle on richer bases, and end-to-end fits in the style of the reference's tests/test_models.py."""
import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise

pytestmark = pytest.mark.gpu


def _imports():
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    return bs, lk, Parameter, Positive, GeneralizedLinearModel


def smse(y_true, y_pred):
    return ((y_true - y_pred) ** 2).mean() / y_true.var()


def _lik(lk, name):
    return {"poisson_exp": lambda: lk.Poisson("exp"), "poisson_softplus": lambda: lk.Poisson("softplus"),
            "gaussian": lk.Gaussian, "bernoulli": lk.Bernoulli, "binomial": lk.Binomial}[name]()


CASES = [("iso", n) for n in ("poisson_exp", "poisson_softplus", "gaussian", "bernoulli", "binomial")] \
    + [("ard", "poisson_exp"), ("ard", "gaussian")]


@pytest.mark.parametrize("tag,lik", CASES)
def test_minibatch_elbo_vs_reference(golden, tag, lik):
    """Same seed -> same standard-normal draws as the reference -> -ELBO and all five gradient blocks."""
    bs, lk, Parameter, Positive, GLM = _imports()
    g = golden("glm")
    X, K, L = g["X"], int(g["K"]), int(g["L"])
    d = X.shape[1]
    ls = g[tag + "_ls"]
    ls = float(ls) if np.ndim(ls) == 0 else ls
    lsp = Parameter(np.ones(d), Positive()) if tag == "ard" else Parameter(1., Positive())
    basis = bs.RandomRBF(nbases=32, Xdim=d, random_state=7, lenscale=lsp)
    assert np.array_equal(basis.W, g["W"])
    glm = GLM(likelihood=_lik(lk, lik), basis=basis, K=K, nsamples=L, random_state=int(g["seed"]))
    glm.B_, glm.D_ = float(g["B"]), 64
    glm._GeneralizedLinearModel__it = -1
    t = tag + "_" + lik
    lp = 0.7 if lik == "gaussian" else []
    largs = (g["nbin"],) if lik == "binomial" else ()
    nobj, (ndm, ndC, dL, dlp, dbp) = glm._elbo(g["m"].copy(), g["C"].copy(), float(g["reg"]), lp, ls, X, g[t + "_y"], *largs)
    glm._release_features()
    assert abs(nobj - g[t + "_obj"]) < 2e-4 * abs(g[t + "_obj"])
    assert normwise(ndm, g[t + "_ndm"]) < 1e-3 and normwise(ndC, g[t + "_ndC"]) < 1e-3
    assert abs(dL - g[t + "_dL"]) < 1e-6 * abs(g[t + "_dL"])
    assert np.shape(dbp) == (() if tag == "iso" else (d,))
    assert normwise(np.atleast_1d(dbp), g[t + "_dbp"]) < 2e-3
    if lik == "gaussian":
        assert normwise(np.atleast_1d(dlp[0]), g[t + "_dlp"]) < 1e-3
    else:
        assert dlp == []


def test_minibatch_elbo_concat_and_generic_children_vs_oracle():
    """RandomMatern32 (ARD) + LinearBasis + FastFoodGM (gradient formed on the host from the downloaded EdPhi
    block) + a second random Fourier basis on a column subset; ragged sizes (M, K*L not multiples of anything)."""
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(3)
    M, d, K, L = 333, 5, 4, 7
    X = rs.randn(M, d)
    y = rs.poisson(np.exp(0.5 * np.sin(X[:, 0]))).astype(float)
    cat = bs.RandomMatern32(nbases=40, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True) \
        + bs.FastFoodGM(nbases=8, Xdim=2, random_state=2, apply_ind=[1, 3]) \
        + bs.RandomRBF(nbases=30, Xdim=2, random_state=4, apply_ind=[0, 2])
    ls0, mu, lsg, ls3 = np.linspace(0.8, 1.3, d), np.array([0.3, -0.2]), np.array([1.1, 0.9]), 0.7
    hyp = [ls0, mu, lsg, ls3]
    Phi = cat.transform(X, *hyp)
    dPs = []
    for gfull in cat.grad(X, *hyp):
        dPs.extend([gfull[:, :, i] for i in range(gfull.shape[2])] if gfull.ndim == 3 else [gfull])
    D = Phi.shape[1]
    m = 0.2 * rs.randn(D, K)
    C = rs.gamma(2., 0.5, size=(D, K))
    regs = [1.2, 0.8, 1.5, 0.6]
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
    assert normwise(np.array(dL), np.array(want[1][2])) < 1e-6
    assert isinstance(dbp, list) and len(dbp) == 4
    assert [np.shape(v) for v in dbp] == [(d,), (2,), (2,), ()]
    flat = np.concatenate([np.atleast_1d(v) for v in dbp])
    assert normwise(flat, np.array(want[1][4])) < 2e-3


@pytest.mark.parametrize("alone", [True, False])
def test_minibatch_elbo_of_a_spectral_mixture_component_on_the_chain_kernel_vs_oracle(alone):
    """FastFoodGM with Xdim > 8 (the chain kernel's mixture mode writes its four blocks, both gradients come from the two
    contractions T+ / T-: basis_functions.py:1443-1537) as the GLM's basis -- ALONE (its gradient is the pair [dmean, dlenscale],
    not one array) and between two other children: -ELBO and every gradient block against the oracle's glm_elbo on this
    basis' own transform / grad, which tests/test_gpu_fastfood.py holds to the reference's."""
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(4)
    M, d, K, L = 257, 12, 3, 5
    X = rs.randn(M, d)
    y = rs.poisson(np.exp(0.5 * np.sin(X[:, 0]))).astype(float)
    gm = bs.FastFoodGM(nbases=24, Xdim=d, random_state=2)
    mu, lsg = 0.3 * rs.randn(d), 0.8 + 0.5 * rs.rand(d)
    if alone:
        basis, hyp, regs = gm, [mu, lsg], 1.3
    else:
        basis = bs.LinearBasis(onescol=True) + gm + bs.RandomRBF(nbases=20, Xdim=d, random_state=4)
        hyp, regs = [mu, lsg, 0.9], [0.8, 1.3, 0.6]
    Phi = basis.transform(X, *hyp)
    dPs = []
    for gfull in (basis.grad(X, *hyp) if not alone else list(basis.grad(X, *hyp))):
        dPs.extend([gfull[:, :, i] for i in range(gfull.shape[2])] if gfull.ndim == 3 else [gfull])
    D = Phi.shape[1]
    m = 0.2 * rs.randn(D, K)
    C = rs.gamma(2., 0.5, size=(D, K))
    Ld, slices = basis.regularizer_diagonal(X, *(regs if isinstance(regs, list) else [regs]))
    e = np.stack([np.random.RandomState(9).randn(K * L, D)[k * L:(k + 1) * L] for k in range(K)])
    want = orc.glm_elbo(m, C, Ld, slices, "poisson_exp", [], (), Phi, dPs, y, e, 4.0)
    glm = GLM(likelihood=lk.Poisson(), basis=basis, K=K, nsamples=L, random_state=9)
    glm.B_, glm.D_ = 4.0, D
    glm._GeneralizedLinearModel__it = -1
    nobj, (ndm, ndC, dL, dlp, dbp) = glm._elbo(m.copy(), C.copy(), regs, [], hyp if not alone else hyp, X, y)
    glm._release_features()
    assert abs(nobj - want[0]) < 2e-4 * abs(want[0])
    assert normwise(ndm, want[1][0]) < 1e-3 and normwise(ndC, want[1][1]) < 1e-3
    assert normwise(np.atleast_1d(np.array(dL)), np.atleast_1d(np.array(want[1][2]))) < 1e-6
    assert isinstance(dbp, list) and [np.shape(v) for v in dbp] == ([(d,), (d,)] if alone else [(d,), (d,), ()])
    flat = np.concatenate([np.atleast_1d(v) for v in dbp])
    assert normwise(flat, np.array(want[1][4])) < 2e-3


def test_minibatch_elbo_on_the_128_tile_products_vs_oracle(monkeypatch):
    """A minibatch of 1280 rows, F = 512, K L = 192: every product of the step (fs = Phi ws, Ed = dfs Phi, EdPhi) is one that
    256 x 256 tiles cannot spread over the CUs and that is too large for the FMA kernel -- rr_gemm_tn_mid_f32_kernel's (128 x 128
    tiles, K-split); the EdPhi product is stored and contracted by rr_glm_grad_t_kernel.  -ELBO and all gradient blocks against
    the oracle's glm_elbo (glm.py:205-294), and against the same evaluation on the tile kernel (RR_GEMM_MID=0 is read once
    per process: the oracle is the reference here, the A/B lives in tools/glm_mid_batch.py)."""
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(5)
    M, d, n, K, L = 1280, 6, 256, 6, 32
    X = rs.randn(M, d)
    y = rs.poisson(np.exp(0.5 * np.sin(X[:, 0]))).astype(float)
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()))
    ls = np.linspace(0.8, 1.4, d)
    Phi = basis.transform(X, ls)
    G = basis.grad(X, ls)
    dPs = [G[:, :, i] for i in range(d)]
    D = 2 * n
    m = 0.2 * rs.randn(D, K)
    C = rs.gamma(2., 0.5, size=(D, K))
    Ld, slices = basis.regularizer_diagonal(X, 1.3)
    e = np.stack([np.random.RandomState(9).randn(K * L, D)[k * L:(k + 1) * L] for k in range(K)])
    want = orc.glm_elbo(m, C, Ld, slices, "poisson_exp", [], (), Phi, dPs, y, e, 3.0)
    glm = GLM(likelihood=lk.Poisson(), basis=basis, K=K, nsamples=L, random_state=9)
    glm.B_, glm.D_ = 3.0, D
    glm._GeneralizedLinearModel__it = -1
    nobj, (ndm, ndC, dL, dlp, dbp) = glm._elbo(m.copy(), C.copy(), 1.3, [], ls, X, y)
    glm._release_features()
    assert abs(nobj - want[0]) < 2e-4 * abs(want[0])
    assert normwise(ndm, want[1][0]) < 1e-3 and normwise(ndC, want[1][1]) < 1e-3
    assert normwise(np.atleast_1d(np.array(dL)), np.atleast_1d(np.array(want[1][2]))) < 1e-6
    assert np.shape(dbp) == (d,) and normwise(np.asarray(dbp), np.concatenate([np.atleast_1d(v) for v in want[1][4]])) < 2e-3


def test_project_and_sample_func():
    bs, lk, Parameter, Positive, GLM = _imports()
    from revrand_amd.basis_functions import MinibatchFeatures
    rs = np.random.RandomState(0)
    X = rs.randn(700, 6)
    basis = bs.RandomRBF(nbases=50, Xdim=6, random_state=1) + bs.LinearBasis(onescol=True)
    W = rs.randn(107, 33)
    f = MinibatchFeatures(basis)
    out = f.project(X, [0.9], W)
    f.release()
    assert normwise(out, basis.transform(X, 0.9) @ W) < 1e-4


def test_glm_gaussian_like_reference_test_models():
    """tests/test_models.py:83-116 of the reference: linear data, LinearBasis then a concatenation; SMSE,
    cdf, logpdf, interval."""
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(100)
    x = np.linspace(-5, 5, 600)
    yall = 3 + 2 * x + rs.randn(600) * 1e-4
    Xall = np.column_stack((np.ones(600), x))
    tr = rs.choice(600, 400, replace=False)
    ts = np.setdiff1d(np.arange(600), tr)
    X, y, Xs, ys = Xall[tr], yall[tr], Xall[ts], yall[ts]
    glm = GLM(lk.Gaussian(), bs.LinearBasis(onescol=True), random_state=1)
    glm.fit(X, y)
    assert smse(ys, glm.predict(Xs)) < 0.1
    basis = bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=20, Xdim=2) + bs.RandomMatern52(nbases=20, Xdim=2)
    glm = GLM(lk.Gaussian(), basis, random_state=1)
    glm.fit(X, y)
    Ey = glm.predict(Xs)
    assert smse(ys, Ey) < 0.1
    py, _, _ = glm.predict_cdf(Xs, 1e5)
    assert np.allclose(py, 1.)
    lpy, _, _ = glm.predict_logpdf(Xs, Ey)
    assert np.all(lpy > -100)
    EyQn, EyQx = glm.predict_interval(Xs[:40], 0.9, multiproc=False)
    assert all(Ey[:40] <= EyQx) and all(Ey[:40] >= EyQn)


def test_glm_binomial_like_reference_test_models():
    """tests/test_models.py:119-147 of the reference."""
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(100)
    x = np.linspace(-50, 50, 600)
    X = x[:, None]
    p = 0.5 * (np.sin(x / 5.) + 1)
    n = 1000
    y = rs.binomial(n, p).astype(float)
    basis = bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=20, Xdim=1) + bs.RandomMatern52(nbases=20, Xdim=1)
    glm = GLM(lk.Binomial(), basis, random_state=1)
    glm.fit(X, y, likelihood_args=(n,))
    Ey = glm.predict(X, likelihood_args=(n,))
    assert smse(p * n, Ey) < 1
    py, _, _ = glm.predict_cdf(X, 1e5, likelihood_args=(n,))
    assert np.allclose(py, 1.)


def test_glm_poisson_large_minibatch():
    """Config 5 in miniature: Poisson, RandomRBF ARD, a large minibatch per step; the fit must improve the
    minibatch objective and recover the rate."""
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(0)
    N, d = 20000, 4
    X = rs.randn(N, d)
    flat = 0.8 * np.sin(1.5 * X[:, 0]) + 0.4 * X[:, 1]
    y = rs.poisson(np.exp(flat)).astype(float)
    basis = bs.RandomRBF(nbases=100, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    glm = GLM(lk.Poisson(), basis, K=3, nsamples=10, batch_size=4096, maxiter=300, nstarts=4, random_state=2)
    glm.fit(X, y)
    Xs = rs.randn(2000, d)
    rate = np.exp(0.8 * np.sin(1.5 * Xs[:, 0]) + 0.4 * Xs[:, 1])
    assert smse(rate, glm.predict(Xs)) < 0.2
    assert glm.weights_.shape == (200, 3) and np.all(glm.covariance_ > 0) and np.shape(glm.basis_hypers_) == (d,)


def test_fit_with_the_library_generator_equals_fit_with_numpy_draws():
    """The reference's random stream from rr_legacy_randn (sequential part on the caller, sqrt / log on a worker thread)
    and from NumPy's RandomState are the same numbers, so a fit is the same fit: identical parameters after 12 SVI
    steps (f32 atomics in the step's reductions allow last-bit differences only), and the RandomState ends in the same
    state either way."""
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(4)
    N, d = 6000, 5
    X = rs.randn(N, d)
    y = rs.poisson(np.exp(0.5 * np.sin(X[:, 0]) + 0.2 * X[:, 2])).astype(float)
    out = []
    for native in (True, False):
        basis = bs.RandomRBF(nbases=64, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
        glm = GLM(lk.Poisson(), basis, K=2, nsamples=7, batch_size=1000, maxiter=12, nstarts=2, random_state=11)
        glm._native_draws = native
        glm.fit(X, y)
        out.append((glm.weights_.copy(), glm.covariance_.copy(), np.array(glm.basis_hypers_, dtype=float), glm.random_.randn()))
    (wa, ca, ha, ra), (wb, cb, hb, rb) = out
    assert normwise(wa, wb) < 1e-5 and normwise(ca, cb) < 1e-5 and normwise(ha, hb) < 1e-5
    assert ra == rb


def test_device_sampler_matches_host_sampler_in_expectation():
    """sampler="device": the reparameterisation draws come from the counter-based generator on the GPU.  With many
    samples the Monte-Carlo gradients agree with the host-sampled ones (same estimator, independent draws), the
    result is reproducible for a seed and changes with the step."""
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(7)
    M, d, n, K, L = 2048, 3, 24, 2, 3000
    X = rs.randn(M, d)
    y = rs.poisson(np.exp(0.4 * np.sin(X[:, 0]))).astype(float)
    D = 2 * n
    m = 0.1 * rs.randn(D, K)
    C = 0.05 * rs.gamma(2., 0.5, size=(D, K))
    out = {}
    for sampler in ("host", "device", "device2"):
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=3)
        glm = GLM(lk.Poisson(), basis, K=K, nsamples=L, random_state=11, sampler=sampler[:6])
        glm.B_, glm.D_ = 4.0, D
        glm._GeneralizedLinearModel__it = -1
        glm._dev_seed = None
        res = [glm._elbo(m.copy(), C.copy(), 1.0, [], 0.9, X, y) for _ in range(2)]
        glm._release_features()
        out[sampler] = res
    h, dv, dv2 = out["host"][0], out["device"][0], out["device2"][0]
    # same seed, same step: the same draws (the split-K GEMM's f32 atomics leave only last-bit differences)
    assert normwise(dv[1][0], dv2[1][0]) < 1e-5 and abs(dv[0] - dv2[0]) < 1e-6 * abs(dv[0])
    assert normwise(out["device"][0][1][0], out["device"][1][1][0]) > 1e-4           # next step: new draws
    # Monte-Carlo agreement, calibrated by the spread between two independent HOST-sampled evaluations
    h2 = out["host"][1]
    assert abs(h[0] - dv[0]) < max(5 * abs(h[0] - h2[0]), 2e-3 * abs(h[0]))        # -ELBO
    for blk in (0, 1):                                                               # -dm, -dC
        noise = normwise(h2[1][blk], h[1][blk])
        assert normwise(dv[1][blk], h[1][blk]) < max(3 * noise, 0.02), (blk, noise)
    assert abs(dv[1][4] - h[1][4]) < max(5 * abs(h2[1][4] - h[1][4]), 0.05 * abs(h[1][4])) + 1e-3   # basis gradient
    with pytest.raises(ValueError):
        GLM(lk.Poisson(), bs.RandomRBF(nbases=4, Xdim=d), sampler="gpu")._elbo(m[:8], C[:8], 1.0, [], 0.9, X, y)


def test_glm_poisson_fit_with_device_sampler():
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(0)
    N, d = 20000, 4
    X = rs.randn(N, d)
    y = rs.poisson(np.exp(0.8 * np.sin(1.5 * X[:, 0]) + 0.4 * X[:, 1])).astype(float)
    basis = bs.RandomRBF(nbases=100, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    glm = GLM(lk.Poisson(), basis, K=3, nsamples=10, batch_size=4096, maxiter=300, nstarts=4, random_state=2,
              sampler="device")
    glm.fit(X, y)
    Xs = rs.randn(2000, d)
    rate = np.exp(0.8 * np.sin(1.5 * Xs[:, 0]) + 0.4 * Xs[:, 1])
    assert smse(rate, glm.predict(Xs)) < 0.2
    from sklearn.base import clone
    assert clone(glm).get_params()["sampler"] == "device"


def test_config5_full_minibatch_is_additive_over_rows():
    """BASELINE config 5's step at full size (M = 65 536 rows, F = 2048, K = 10, L = 50, Poisson, ARD), through a
    size-independent property instead of the CPU oracle (minutes at this size): for fixed weight samples every
    output of the step is a sum over rows, so the full minibatch equals the sum of its two halves; and a 1024-row
    slice agrees with the oracle."""
    bs, lk, Parameter, Positive, GLM = _imports()
    from revrand_amd.basis_functions import MinibatchFeatures
    rs = np.random.RandomState(5)
    M, d, n, K, L = 65536, 32, 1024, 10, 50
    X = rs.randn(M, d).astype(np.float32).astype(np.float64)
    y = rs.poisson(np.exp(0.4 * np.sin(X[:, 0]))).astype(float)
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    ls = np.linspace(0.8, 1.4, d)
    WS = 0.05 * rs.randn(K * L, 2 * n)
    f = MinibatchFeatures(basis)

    def step(sl):
        f.assemble(X[sl], [ls])
        Edws, ll, aux = f.glm_step(y[sl], None, lk.RR_LIK_POISSON_EXP, 0.0, WS, K, L)
        return Edws, ll, np.asarray(f.glm_basis_grads(X[sl]))

    full, a, b = step(slice(0, M)), step(slice(0, M // 2)), step(slice(M // 2, M))
    assert normwise(a[0] + b[0], full[0]) < 1e-4 and normwise(a[1] + b[1], full[1]) < 1e-5
    assert normwise(a[2] + b[2], full[2]) < 1e-3
    # a slice against the oracle's formulas (glm.py:296-322 with the same weight samples)
    sl = slice(1000, 2024)
    Edws, ll, g = step(sl)
    f.release()
    Phi = orc.rff_transform(X[sl], basis.W, ls)
    fs = WS @ Phi.T
    dfs = orc.lik_df("poisson_exp", y[sl], fs)
    assert normwise(Edws, dfs @ Phi) < 1e-3
    EdPhi = dfs.T @ WS / (K * L)
    dP = orc.rff_grad(X[sl], basis.W, ls)
    assert normwise(g, np.array([-(EdPhi * dP[:, :, i]).sum() for i in range(d)])) < 5e-3


@pytest.mark.parametrize("d,n,rows,ard", [(5, 256, 1500, True), (5, 256, 1500, False), (32, 512, 700, True),
                                          (17, 256, 256, True), (3, 768, 2049, False), (40, 256, 600, True),
                                          (64, 256, 300, True), (100, 256, 777, True)])
def test_edphi_product_fused_with_its_contraction_equals_the_two_pass_route(monkeypatch, d, n, rows, ard):
    """A lone random Fourier child whose [cos | sin] block fills whole 256-column tiles: the step contracts every block
    of EdPhi = dfs^T ws / (K L) (glm.py:311) with P and X while it is in registers (rr_featmat_glm_plan_rff /
    rr_gemm_gradt_f32_kernel) instead of writing EdPhi and reading it back.  Same length-scale gradients as the two-pass
    route (RR_GLM_NO_FUSE=1: GEMM, then rr_glm_grad_t_kernel) and as the oracle's -(EdPhi o dPhi_i).sum() (glm.py:274-275),
    with partial last row tiles, d below 32 and above (two / four 32-column blocks of X), isotropic (the reference's
    dimension-0 quirk) and ARD length scales, on the host-sample route and on both reduced routes."""
    bs, lk, Parameter, Positive, GLM = _imports()
    from revrand_amd.basis_functions import MinibatchFeatures
    rs = np.random.RandomState(100 + d + n)
    K, L = 3, 7
    X = rs.randn(rows, d).astype(np.float32).astype(np.float64)
    y = rs.poisson(np.exp(0.4 * np.sin(X[:, 0]))).astype(float)
    if ard:
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
        ls = np.linspace(0.8, 1.4, d)
    else:
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1)
        ls = 1.1
    WS = 0.05 * rs.randn(K * L, 2 * n)
    m, C = 0.05 * rs.randn(2 * n, K), 0.01 + 0.01 * rs.rand(2 * n, K)
    E = rs.randn(K * L, 2 * n).astype(np.float32)

    def run():
        f = MinibatchFeatures(basis)
        out = []
        for route in ("samples", "draws", "sampled"):
            f.assemble(X, [ls])
            if route == "samples":
                f.glm_step(y, None, lk.RR_LIK_POISSON_EXP, 0.0, WS, K, L)
            elif route == "draws":
                f.glm_step_draws(y, None, lk.RR_LIK_POISSON_EXP, 0.0, m, C, K, L, E)
            else:
                f.glm_step_sampled(y, None, lk.RR_LIK_POISSON_EXP, 0.0, m, C, K, L, 7, 3)
            out.append(np.atleast_1d(np.asarray(f.glm_basis_grads(X), dtype=float)))
        # an objective-only step in between leaves no plan behind for the next one
        f.assemble(X, [ls])
        f.glm_step_draws(y, None, lk.RR_LIK_POISSON_EXP, 0.0, m, C, K, L, E, objective_only=True)
        f.assemble(X, [ls])
        f.glm_step(y, None, lk.RR_LIK_POISSON_EXP, 0.0, WS, K, L)
        out.append(np.atleast_1d(np.asarray(f.glm_basis_grads(X), dtype=float)))
        f.release()
        return out

    monkeypatch.delenv("RR_GLM_NO_FUSE", raising=False)
    fused = run()
    monkeypatch.setenv("RR_GLM_NO_FUSE", "1")
    plain = run()
    for a, b in zip(fused, plain):
        assert a.shape == b.shape and normwise(a, b) < 2e-4
    assert normwise(fused[3], fused[0]) < 1e-5
    # the host-sample route against the oracle's formulas
    Phi = orc.rff_transform(X, basis.W, ls)
    dfs = orc.lik_df("poisson_exp", y, WS @ Phi.T)
    EdPhi = dfs.T @ WS / (K * L)
    dP = orc.rff_grad(X, basis.W, ls)
    want = np.array([-(EdPhi * dP[:, :, i]).sum() for i in range(d)]) if ard else np.array([-(EdPhi * dP).sum()])
    assert normwise(fused[0], want) < 5e-3


@pytest.mark.parametrize("likname", ["poisson_exp", "gaussian", "bernoulli", "binomial", "poisson_softplus"])
@pytest.mark.parametrize("rows,n,K,L", [(700, 100, 3, 7), (1024, 128, 2, 150), (300, 40, 1, 5)])
def test_first_product_with_the_likelihood_terms_as_its_epilogue_equals_the_three_pass_route(monkeypatch, likname, rows, n, K, L):
    """fs = Phi ws^T, the likelihood derivatives and dfs^T as ONE kernel (rr_gemm_lik_f32_kernel: the 256x256 block of fs
    turns into dfs in registers and is stored in both layouts) against GEMM + rr_glm_lik_kernel + transposing pass
    (RR_GLM_FUSE_LIK=0): every output of the step (glm.py:296-322), the objective-only evaluation, partial row tiles,
    K L below / above one 256-column tile.  All five likelihoods (the logistic / softplus ones through rr_log1p01).  (The
    fused route is taken by itself from 2 x CU-count output tiles on -- config 5 -- and forced here.)"""
    bs, lk, Parameter, Positive, GLM = _imports()
    from revrand_amd.basis_functions import MinibatchFeatures
    rs = np.random.RandomState(rows + n)
    d = 6
    X = rs.randn(rows, d)
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    ls = np.linspace(0.8, 1.4, d)
    rowarg, oargs = None, []
    if likname == "gaussian":
        y, lid, lpar, oargs = np.sin(X[:, 0]) + 0.1 * rs.randn(rows), lk.RR_LIK_GAUSSIAN, 0.3, [0.3]
    elif likname == "bernoulli":
        y, lid, lpar = (rs.rand(rows) < 0.5 + 0.3 * np.sin(X[:, 0])).astype(float), lk.RR_LIK_BERNOULLI, 0.0
    elif likname == "binomial":
        rowarg = rs.randint(1, 9, size=rows).astype(float)
        y, lid, lpar, oargs = rs.binomial(rowarg.astype(int), 0.5 + 0.3 * np.sin(X[:, 0])).astype(float), lk.RR_LIK_BINOMIAL, 0.0, [rowarg]
    else:
        y = rs.poisson(np.exp(0.4 * np.sin(X[:, 0]))).astype(float)
        lid, lpar = (lk.RR_LIK_POISSON_EXP if likname == "poisson_exp" else lk.RR_LIK_POISSON_SOFTPLUS), 0.0
    WS = 0.05 * rs.randn(K * L, 2 * n)
    m, C = 0.05 * rs.randn(2 * n, K), 0.01 + 0.01 * rs.rand(2 * n, K)
    E = rs.randn(K * L, 2 * n).astype(np.float32)

    def run():
        f = MinibatchFeatures(basis)
        f.assemble(X, [ls])
        a = f.glm_step(y, rowarg, lid, lpar, WS, K, L) + (np.asarray(f.glm_basis_grads(X)),)
        f.assemble(X, [ls])
        b = f.glm_step_draws(y, rowarg, lid, lpar, m, C, K, L, E) + (np.asarray(f.glm_basis_grads(X)),)
        f.assemble(X, [ls])
        c = f.glm_step_draws(y, rowarg, lid, lpar, m, C, K, L, E, objective_only=True)[2:]
        f.release()
        return list(a) + list(b) + list(c)

    monkeypatch.setenv("RR_GLM_FUSE_LIK", "force")
    fused = run()
    monkeypatch.setenv("RR_GLM_FUSE_LIK", "0")
    plain = run()
    for u, v in zip(fused, plain):
        assert np.shape(u) == np.shape(v)
        assert normwise(u, v) < 2e-5 or (np.abs(np.asarray(u) - np.asarray(v)).max() < 1e-9)
    # and against the oracle's formulas for the host-sample route
    Phi = orc.rff_transform(X, basis.W, ls)
    fs = WS @ Phi.T
    dfs = orc.lik_df(likname, y, fs, *oargs)
    assert normwise(fused[0], dfs @ Phi) < 1e-3


def test_step_with_directly_written_transpose_equals_step_with_transposing_pass():
    """From the second step of a given minibatch size on, the random Fourier children write their blocks of P^T while they
    write P and the step skips its transposing pass: same step results as the first (transposing) step on the same
    inputs, also after a step of another size in between (which falls back and lays the padding out again), and for a
    concatenation with a linear child (never covered: always the transposing pass)."""
    bs, lk, Parameter, Positive, GLM = _imports()
    from revrand_amd.basis_functions import MinibatchFeatures
    rs = np.random.RandomState(12)
    N, d, K, L = 1500, 7, 2, 6
    X = rs.randn(N, d)
    y = rs.poisson(np.exp(0.4 * X[:, 1])).astype(float)
    for basis, hyp in [(bs.RandomRBF(nbases=100, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive())),
                        [np.linspace(0.7, 1.2, d)]),
                       (bs.RandomRBF(nbases=70, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()))
                        + bs.RandomMatern32(nbases=33, Xdim=d, random_state=4), [np.linspace(0.7, 1.2, d), 1.1]),
                       (bs.RandomRBF(nbases=40, Xdim=d, random_state=3) + bs.LinearBasis(onescol=True), [0.9])]:
        D = int(basis.get_dim(X))
        WS = 0.1 * rs.randn(K * L, D)
        f = MinibatchFeatures(basis)
        out = []
        for rows in (1000, 1000, 1000, 777, 1000, 1000):
            f.assemble(X[:rows], hyp)
            E, ll, _ = f.glm_step(y[:rows], None, lk.RR_LIK_POISSON_EXP, 0.0, WS, K, L)
            g = f.glm_basis_grads(X[:rows])
            out.append((rows, E.copy(), np.array(ll), np.concatenate([np.atleast_1d(np.asarray(v, dtype=float)).ravel()
                                                                       for v in (g if isinstance(g, list) else [g])])))
        f.release()
        ref = out[0]
        for rows, E, ll, g in out[1:]:
            if rows == ref[0]:
                assert normwise(E, ref[1]) < 1e-5 and normwise(ll, ref[2]) < 1e-6
                if g.size:
                    assert normwise(g, ref[3]) < 1e-4


def test_resident_minibatch_gather_equals_host_gather():
    """fit() keeps X on the device and gathers minibatches there by index: the step on rows `idx` of the resident data
    equals the step on the host-gathered X[idx] (concatenation with Linear + Bias, column subsets)."""
    bs, lk, Parameter, Positive, GLM = _imports()
    from revrand_amd.basis_functions import MinibatchFeatures
    rs = np.random.RandomState(2)
    N, d, K, L = 5000, 6, 3, 5
    X = rs.randn(N, d)
    y = rs.poisson(np.exp(0.3 * X[:, 0])).astype(float)
    cat = bs.RandomRBF(nbases=30, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True, apply_ind=[1, 4]) + bs.BiasBasis(offset=0.5) \
        + bs.RandomMatern32(nbases=10, Xdim=2, random_state=2, apply_ind=[0, 5])
    hyp = [np.linspace(0.8, 1.2, d), 0.9]
    D = int(cat.get_dim(X))
    WS = 0.1 * rs.randn(K * L, D)
    idx = rs.permutation(N)[:777]
    a = MinibatchFeatures(cat)
    assert a.make_resident(X)
    a.assemble_idx(idx, hyp)
    Ea, la, _ = a.glm_step(y[idx], None, lk.RR_LIK_POISSON_EXP, 0.0, WS, K, L)
    ga = a.glm_basis_grads(np.empty((777, 0)))
    a.release()
    b = MinibatchFeatures(cat)
    b.assemble(X[idx], hyp)
    Eb, lb, _ = b.glm_step(y[idx], None, lk.RR_LIK_POISSON_EXP, 0.0, WS, K, L)
    gb = b.glm_basis_grads(X[idx])
    b.release()
    assert normwise(Ea, Eb) < 1e-5 and normwise(la, lb) < 1e-6
    assert normwise(np.concatenate([np.atleast_1d(v) for v in ga]), np.concatenate([np.atleast_1d(v) for v in gb])) < 1e-4
    # a basis that cannot stay resident (f64 arithmetic) declines, and nothing is kept
    c = MinibatchFeatures(bs.RandomRBF(nbases=8, Xdim=d, dtype="f64") + bs.LinearBasis())
    assert not c.make_resident(X)


def test_minibatches_gathered_ahead_on_the_worker_equal_minibatches_gathered_by_the_step(monkeypatch):
    """with the data resident, the minibatch worker uploads a future step's row indices and targets and
    gathers its rows on a second stream while the current step runs (three buffer sets in turn).  Same batches, same
    draws: the fit equals the fit whose steps do all of that themselves (RR_GLM_BATCH_PREFETCH=0), to the rounding of the
    step's atomic sums; binomial (a per-row likelihood argument) and Gaussian (parameter-dependent spec) included."""
    bs, lk, Parameter, Positive, GLM = _imports()
    from revrand_amd.basis_functions import MinibatchFeatures
    rs = np.random.RandomState(5)
    N, d = 6000, 5
    X = rs.randn(N, d)
    f = 0.8 * np.sin(X[:, 0]) + 0.3 * X[:, 1]
    cases = [(lk.Poisson(), rs.poisson(np.exp(f)).astype(float), ()),
             (lk.Binomial(), rs.binomial(7, 1 / (1 + np.exp(-f))).astype(float), (7 * np.ones(N),)),
             (lk.Gaussian(), f + 0.1 * rs.randn(N), ())]
    calls = []
    orig = MinibatchFeatures.prefetch_batch

    def counting(self, updev, idx, y, rowarg):
        calls.append(len(idx))
        return orig(self, updev, idx, y, rowarg)
    monkeypatch.setattr(MinibatchFeatures, "prefetch_batch", counting)
    for lik, y, largs in cases:
        fits = []
        for pf in ("1", "0"):
            monkeypatch.setenv("RR_GLM_BATCH_PREFETCH", pf)
            calls.clear()
            basis = bs.RandomRBF(nbases=40, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) \
                + bs.LinearBasis(onescol=True)
            glm = GLM(lik, basis, K=2, nsamples=8, batch_size=500, maxiter=12, nstarts=2, random_state=3)
            glm.fit(X, y, likelihood_args=largs)
            assert (len(calls) > 10) == (pf == "1"), (pf, len(calls))
            fits.append(np.concatenate((glm.weights_.ravel(), glm.covariance_.ravel(),
                                        np.concatenate([np.atleast_1d(h) for h in glm.basis_hypers_ if np.size(h)]))))
        assert normwise(fits[0], fits[1]) < 1e-5, type(lik).__name__


def _gloo_gpu_glm_worker(rank, world, port, q):
    """One rank of the row-sharded SVI with the REAL device features (both ranks share GPU 0); gloo carries the one
    all-reduce per step on the host.  Mirrors tests/test_dist_gloo.py::_glm_worker without its NumPy stand-in."""
    import os
    import torch.distributed as dist
    from revrand_amd import parallel
    from revrand_amd.optimize import Adam
    bs, lk, Parameter, Positive, GLM = _imports()
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(0)
        N, d, n, K, L = 2400, 3, 40, 2, 6
        X = rs.randn(N, d)
        y = rs.poisson(np.exp(0.3 * np.sin(X[:, 0]))).astype(float)
        a, b = parallel.shard_bounds(N, rank, world)
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=5, lenscale=Parameter(np.ones(d), Positive()),
                             regularizer=Parameter(1.5, Positive()))
        glm = GLM(lk.Poisson(), basis, K=K, nsamples=L, batch_size=b - a, maxiter=6, nstarts=0, random_state=3,
                  updater=Adam(alpha=0.05), distributed=world > 1)
        glm.B_, glm.D_ = 1.0, 2 * n
        glm._GeneralizedLinearModel__it = -1
        m, C = 0.1 * rs.randn(2 * n, K), rs.gamma(2., 0.5, (2 * n, K))
        f, (ndm, ndC, dL, dlp, dbp) = glm._elbo(m, C, 1.5, [], np.array([0.9, 1.1, 1.3]), X[a:b], y[a:b])
        ev = [float(f), float(dL)] + ndm.ravel().tolist() + ndC.ravel().tolist() + np.asarray(dbp).tolist()
        glm._release_features()
        glm.random_ = np.random.RandomState(3)
        glm.fit(X[a:b], y[a:b])
        q.put((rank, ev, glm.weights_.ravel().tolist(), np.asarray(glm.basis_hypers_).tolist(), float(glm.regularizer_)))
    finally:
        if world > 1:
            dist.destroy_process_group()


def _join_or_kill(procs, timeout=120):
    """Every worker must have exited cleanly within `timeout` s of delivering its result; one that has not is ended (it
    would otherwise be joined forever by multiprocessing's exit handler) and fails the test."""
    codes = []
    for p in procs:
        p.join(timeout)
        if p.is_alive():
            p.kill()
            p.join(10)
            codes.append("still running %d s after its result (killed)" % timeout)
        else:
            codes.append(p.exitcode)
    assert all(c == 0 for c in codes), codes


@pytest.mark.timeout(1600)  # worst case: two 600 s waits on the result queue + three 120 s joins
def test_two_rank_gloo_glm_with_real_device_features():
    """Row-sharded SVI on two processes with real kernels: every rank's minibatch covers its shard, so the all-reduced
    `_elbo` equals the single-process evaluation on all rows (same seed -> same draws) to f32 accuracy, and after a short
    distributed `fit` both ranks hold identical parameters."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_gpu_glm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=600) for _ in procs])
    finally:
        _join_or_kill(procs)
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_gloo_gpu_glm_worker, args=(0, 1, 0, q1))
    p1.start()
    try:
        single = q1.get(timeout=600)
    finally:
        _join_or_kill([p1])
    assert res[0][1] == res[1][1]                                                # ranks agree exactly
    ref = np.array(single[1])
    assert np.abs(np.array(res[0][1]) - ref).max() < 2e-4 * np.abs(ref).max()    # and equal the all-rows evaluation
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] and res[0][4] == res[1][4]


@pytest.mark.parametrize("sampler", ["host", "device"])
def test_objective_only_step_equals_full_objective(sampler):
    """The random starts of `fit` rank candidates by -ELBO alone: the objective-only step (fs + likelihood sums, no
    gradient GEMMs) returns the objective of the full step for the same draws, and a fit with random starts runs."""
    bs, lk, Parameter, Positive, GLM = _imports()
    rs = np.random.RandomState(2)
    M, d, n, K, L = 3000, 4, 60, 3, 7
    X = rs.randn(M, d)
    y = rs.poisson(np.exp(0.4 * np.sin(X[:, 0]))).astype(float)
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True)
    D = 2 * n + d + 1
    m, C = 0.1 * rs.randn(D, K), rs.gamma(2., 0.5, (D, K))
    vals = []
    for oo in (False, True):
        glm = GLM(lk.Poisson(), basis, K=K, nsamples=L, random_state=11, sampler=sampler)
        glm.B_, glm.D_ = 4.0, D
        glm._GeneralizedLinearModel__it = -1
        if sampler == "device":
            glm._dev_seed, glm._dev_step = 12345, 3
        r = glm._elbo(m, C, [1.2, 0.7], [], [np.linspace(0.8, 1.2, d)], X, y, objective_only=oo)
        glm._release_features()
        vals.append(r if oo else r[0])
    assert abs(vals[0] - vals[1]) < 1e-9 * abs(vals[0])
    glm = GLM(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1), K=K, nsamples=L, random_state=5, sampler=sampler,
              nstarts=6, maxiter=20, batch_size=500)
    glm.fit(X, y)
    assert np.isfinite(glm.predict(X[:20])).all()


@pytest.mark.parametrize("lik", ["gaussian", "poisson_exp", "binomial"])
def test_prediction_surface_vs_reference(golden, lik):
    """`_sample_func`, `predict_moments`, `predict_logpdf`, `predict_cdf`, `predict_interval` of a model whose fitted attributes
    are set, `random_` seeded as the reference's was (tests/golden/glm_predict.npz, oracle/make_golden.py: gen_glm_predict):
    the reference's draws (randint then randn, glm.py:606-610), the latent samples f = Phi w from the device
    (rr_featmat_project), the likelihoods' Ey / loglike / cdf, and the intervals -- the reference root-finds per row with
    brentq (glm.py:659-690), here every row is bisected at once: the same quantiles (for the count likelihoods the sampled CDF
    is a step function and both land on the step)."""
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd.glm import GeneralizedLinearModel
    g = golden("glm_predict")
    X, K, S = g["X"], int(g["K"]), int(g["S"])
    d = X.shape[1]
    basis = bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=g["W"].shape[1], Xdim=d, random_state=8)
    assert np.array_equal(basis.bases[1].W, g["W"])
    like = {"gaussian": lk.Gaussian, "poisson_exp": lambda: lk.Poisson("exp"), "binomial": lk.Binomial}[lik]()
    glm = GeneralizedLinearModel(like, basis, K=K, random_state=0)
    glm.weights_, glm.covariance_, glm.regularizer_ = g["m"], g["C"], [1.0, 1.0]
    glm.like_hypers_ = 0.3 if lik == "gaussian" else []
    glm.basis_hypers_ = float(g["ls"])
    largs = (g["nbin"],) if lik == "binomial" else ()

    def seeded(fn, *a, **k):
        glm.random_ = np.random.RandomState(77)
        return fn(*a, **k)
    fs = np.array(list(seeded(glm._sample_func, X, S)))
    assert normwise(fs, g[lik + "_fs"]) < 1e-5          # (float32 features on the device)
    rows = np.array(list(seeded(glm._sample_func, X, S, genaxis=0)))
    assert normwise(rows, g[lik + "_fs"].T) < 1e-5
    Ey, Vy = seeded(glm.predict_moments, X, S, likelihood_args=largs)
    assert normwise(Ey, g[lik + "_Ey"]) < 1e-5 and normwise(Vy, g[lik + "_Vy"]) < 1e-4
    assert normwise(seeded(glm.predict, X, S, likelihood_args=largs), g[lik + "_Ey"]) < 1e-5
    lp = seeded(glm.predict_logpdf, X, g["yq_" + lik], S, likelihood_args=largs)
    assert normwise(np.array(lp), g[lik + "_logpdf"]) < 1e-5
    cdf = seeded(glm.predict_cdf, X, float(g[lik + "_q"]), S, likelihood_args=largs)
    assert normwise(np.array(cdf), g[lik + "_cdf"]) < 1e-5
    ql, qu = seeded(glm.predict_interval, X[:12], 0.9, S, likelihood_args=tuple(a[:12] for a in largs))
    tol = 1e-4 if lik == "gaussian" else 1e-6
    assert np.all(np.abs(ql - g[lik + "_ql"]) < tol * np.maximum(1.0, np.abs(g[lik + "_ql"])))
    assert np.all(np.abs(qu - g[lik + "_qu"]) < tol * np.maximum(1.0, np.abs(g[lik + "_qu"])))
