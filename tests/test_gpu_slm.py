"""
GPU parity of StandardLinearModel (reference: revrand/slm.py) -- single `_elbo` evaluations
against the golden vectors the reference produced (ELBO, every gradient, posterior m and C),
an end-to-end fit, and the sklearn-facing behaviour the reference's tests/test_models.py covers.
"""
import numpy as np
import pytest
from sklearn.base import clone
from sklearn.decomposition import PCA
from sklearn.model_selection import GridSearchCV
from sklearn.pipeline import Pipeline

import revrand_oracle as orc
from conftest import ROOT, normwise

pytestmark = pytest.mark.gpu


def smse(y_true, y_pred):
    return ((y_true - y_pred) ** 2).sum() / (len(y_true) * y_true.var())


def _imports():
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    return bs, Parameter, Positive, StandardLinearModel


def test_dense_gram_vs_numpy():
    from revrand_amd import _hip
    rs = np.random.RandomState(0)
    for (N, F) in [(1, 1), (33, 7), (700, 300), (5000, 513)]:
        Phi = rs.randn(N, F)
        y = rs.randn(N)
        G, b, yty = _hip.dense_gram(Phi, y)
        assert np.array_equal(G, G.T)
        assert normwise(G, Phi.T @ Phi) < 1e-5
        assert normwise(b, Phi.T @ y) < 1e-5
        assert abs(yty - y @ y) < 1e-10 * max(1.0, y @ y)
    G, b, yty = _hip.dense_gram(rs.randn(40, 5).astype(np.float32))
    assert b is None and G.shape == (5, 5)


@pytest.mark.parametrize("tag", ["iso", "ard"])
def test_elbo_single_basis_vs_reference(golden, tag):
    bs, Parameter, Positive, SLM = _imports()
    g = golden("elbo")
    X, y = g["X"], g["y"]
    d, n = X.shape[1], 16
    lsp = Parameter(1., Positive()) if tag == "iso" else Parameter(np.ones(d), Positive())
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=21, lenscale=lsp)
    assert np.array_equal(basis.W, g[tag + "_W"])
    slm = SLM(basis)
    slm.obj_ = -np.inf
    ls = float(g[tag + "_ls"]) if tag == "iso" else g[tag + "_ls"]
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, float(g["var"]), float(g["reg"]), ls)
    assert abs(-nelbo - g[tag + "_elbo"]) < 1e-4 * abs(g[tag + "_elbo"])
    assert normwise(slm.weights_, g[tag + "_m"]) < 1e-3
    assert normwise(slm.covariance_, g[tag + "_C"]) < 1e-3
    assert normwise(-ndvar, g[tag + "_dvar"]) < 1e-3
    assert normwise(-np.atleast_1d(ndreg), g[tag + "_dreg"]) < 1e-3
    assert normwise(-np.atleast_1d(ndhyp), g[tag + "_dhyp"]) < 2e-3


def test_elbo_concatenation_vs_reference(golden):
    bs, Parameter, Positive, SLM = _imports()
    g = golden("elbo")
    X, y = g["X"], g["y"]
    d, n = X.shape[1], 16
    basis = bs.RandomMatern52(nbases=n, Xdim=d, random_state=22,
                              lenscale=Parameter(np.ones(d), Positive())) + bs.LinearBasis(onescol=True)
    assert np.array_equal(basis.bases[0].W, g["cat_W"])
    slm = SLM(basis)
    slm.obj_ = -np.inf
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, float(g["var"]), list(g["cat_reg"]), g["cat_ls"])
    assert abs(-nelbo - g["cat_elbo"]) < 1e-4 * abs(g["cat_elbo"])
    assert normwise(slm.weights_, g["cat_m"]) < 1e-3
    assert normwise(slm.covariance_, g["cat_C"]) < 1e-3
    assert normwise(-np.asarray(ndreg), g["cat_dreg"]) < 1e-3
    assert normwise(-np.atleast_1d(ndhyp), g["cat_dhyp"]) < 2e-3


def test_fit_end_to_end_vs_reference(golden):
    """Same data, W, fixed initial values, nstarts=0, maxiter=20 as the reference run."""
    bs, Parameter, Positive, SLM = _imports()
    g = golden("fit")
    X, y, Xs = g["X"], g["y"], g["Xs"]
    basis = bs.RandomRBF(nbases=24, Xdim=3, random_state=31, lenscale=Parameter(1.2, Positive()),
                         regularizer=Parameter(1.5, Positive()))
    assert np.array_equal(basis.W, g["W"])
    slm = SLM(basis, var=Parameter(0.5, Positive()), nstarts=0, maxiter=20, random_state=0).fit(X, y)
    Ey, Vy = slm.predict_moments(Xs)
    # L-BFGS trajectories are sensitive: compare at prediction level
    assert smse(g["Ey"], Ey) < 1e-3
    assert np.all(Vy > 0) and normwise(Vy, g["Vy"]) < 0.2
    assert abs(slm.obj_ - float(g["obj"])) < 0.02 * abs(float(g["obj"]))
    # predictions are consistent with the oracle given the fitted state
    Phi = orc.rff_transform(Xs, basis.W, slm.hypers_)
    Eo, Vo = orc.slm_predict_moments(Phi, slm.weights_, slm.covariance_, slm.var_)
    assert normwise(Ey, Eo) < 1e-3 and normwise(Vy, Vo) < 1e-3


def _gaus_data():
    rs = np.random.RandomState(99)
    x = np.linspace(-5, 5, 600)
    y = 3 + 2 * x + rs.randn(600) * 1e-4
    X = np.hstack((np.ones((600, 1)), x[:, None]))
    tr = rs.choice(600, 400, replace=False)
    mask = np.zeros(600, dtype=bool)
    mask[tr] = True
    return X[mask], y[mask], X[~mask], y[~mask]


def test_slm_like_reference_test_models():
    """tests/test_models.py:16-36 of the reference: smse < 0.1 for linear and concatenated bases."""
    bs, Parameter, Positive, SLM = _imports()
    X, y, Xs, ys = _gaus_data()
    slm = SLM(bs.LinearBasis(onescol=False), nstarts=10, random_state=1).fit(X, y)
    assert smse(ys, slm.predict(Xs)) < 0.1
    basis = bs.LinearBasis(onescol=False) + bs.RandomRBF(nbases=10, Xdim=2, random_state=2) \
        + bs.RandomMatern52(nbases=10, Xdim=2, random_state=3)
    slm = SLM(basis, nstarts=10, random_state=1, maxiter=50).fit(X, y)
    assert smse(ys, slm.predict(Xs)) < 0.1


def test_sklearn_protocol():
    """Pipeline / GridSearchCV / clone (tests/test_models.py:39-80,197-238), single process."""
    bs, Parameter, Positive, SLM = _imports()
    X, y, Xs, ys = _gaus_data()
    slm = SLM(bs.LinearBasis(onescol=True), nstarts=0)
    pipe = Pipeline([("PCA", PCA()), ("SLM", slm)]).fit(X, y)
    assert smse(ys, pipe.predict(Xs)) < 0.1
    est = GridSearchCV(slm, {"var": [Parameter(v, Positive()) for v in [1.0, 2.0]]}, cv=2).fit(X, y)
    assert len(est.predict(Xs)) == len(ys)
    c = clone(slm)
    assert repr(c.get_params()["basis"]) == repr(slm.basis) and not hasattr(c, "weights_")
    from sklearn.exceptions import NotFittedError
    with pytest.raises(NotFittedError):
        c.predict(Xs)


@pytest.mark.parametrize("tag", ["iso", "ard"])
def test_elbo_f64_mode_vs_reference(golden, tag):
    """dtype='f64': the whole `_elbo` within 1e-5 of the reference (BASELINE fp64 tolerance).
    (The hyper-gradient's cross-Gram still runs through the f32 SYRK: 1e-3 there.)"""
    bs, Parameter, Positive, SLM = _imports()
    g = golden("elbo")
    X, y = g["X"], g["y"]
    d, n = X.shape[1], 16
    lsp = Parameter(1., Positive()) if tag == "iso" else Parameter(np.ones(d), Positive())
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=21, lenscale=lsp, dtype="f64")
    slm = SLM(basis)
    slm.obj_ = -np.inf
    ls = float(g[tag + "_ls"]) if tag == "iso" else g[tag + "_ls"]
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, float(g["var"]), float(g["reg"]), ls)
    assert abs(-nelbo - g[tag + "_elbo"]) < 1e-9 * abs(g[tag + "_elbo"])
    assert normwise(slm.weights_, g[tag + "_m"]) < 1e-5
    assert normwise(slm.covariance_, g[tag + "_C"]) < 1e-5
    assert normwise(-ndvar, g[tag + "_dvar"]) < 1e-5
    assert normwise(-np.atleast_1d(ndreg), g[tag + "_dreg"]) < 1e-5
    assert normwise(-np.atleast_1d(ndhyp), g[tag + "_dhyp"]) < 2e-3


@pytest.mark.parametrize("tag", ["iso", "ard"])
def test_resident_elbo_matches_reference(golden, tag):
    """The device-resident `_elbo` (no Phi, no dPhi: statistics pass + U = Phi C GEMM pass) against the
    reference's golden ELBO / gradients / posterior."""
    bs, Parameter, Positive, SLM = _imports()
    g = golden("elbo")
    X, y = g["X"], g["y"]
    d, n = X.shape[1], 16
    lsp = Parameter(1., Positive()) if tag == "iso" else Parameter(np.ones(d), Positive())
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=21, lenscale=lsp)
    slm = SLM(basis)
    slm.obj_ = -np.inf
    slm._state = basis.device_fit_state(X, y)
    assert slm._state is not None
    ls = float(g[tag + "_ls"]) if tag == "iso" else g[tag + "_ls"]
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, float(g["var"]), float(g["reg"]), ls)
    slm._state.release()
    assert abs(-nelbo - g[tag + "_elbo"]) < 1e-4 * abs(g[tag + "_elbo"])
    assert normwise(slm.weights_, g[tag + "_m"]) < 1e-3
    assert normwise(-ndvar, g[tag + "_dvar"]) < 1e-3
    assert normwise(-np.atleast_1d(ndreg), g[tag + "_dreg"]) < 1e-3
    assert normwise(-np.atleast_1d(ndhyp), g[tag + "_dhyp"]) < 2e-3
    assert np.shape(ndhyp) == (() if tag == "iso" else (d,))


@pytest.mark.parametrize("shape", [(700, 5, 40), (1500, 16, 96), (1025, 21, 130)])
def test_second_pass_and_predict_vs_oracle(shape):
    """sqErr, the gradient contraction and predict_moments on the device vs the oracle's dPhi-based formulas."""
    bs, Parameter, Positive, SLM = _imports()
    N, d, n = shape
    rs = np.random.RandomState(N)
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    basis = bs.RandomMatern52(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()))
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
    assert normwise(dh, -np.array(o["dhyp"])) < 2e-3        # reference passes -dELBO/dl to the optimiser
    Xs = rs.randn(257, d)
    Ey, Vf = basis.predict_moments(Xs, ls, o["m"], o["C"])
    Eo, Vo = orc.slm_predict_moments(orc.rff_transform(Xs, basis.W, ls), o["m"], o["C"], 0.0)
    assert normwise(Ey, Eo) < 1e-4 and normwise(Vf, Vo) < 1e-3


@pytest.mark.parametrize("N,d,n,ard,chunk", [(3000, 5, 256, True, None), (1500, 32, 256, False, None), (2100, 7, 512, True, 768),
                                             (1500, 64, 256, True, None), (900, 100, 256, True, 512)])
def test_second_pass_product_fused_with_its_contraction_equals_the_two_pass_route(monkeypatch, N, d, n, ard, chunk):
    """With the [cos | sin] block in whole 256-column tiles the second pass' U = Phi C (slm.py:193-195) contracts itself
    with Phi, Err m^T and X block by block in registers (rr_gemm_gradt_f32_kernel<true>) -- U is neither stored nor read
    back.  Same sqErr and hyper-gradient as the GEMM + rr_grad_t_kernel route (RR_PASS2_NO_FUSE=1) and as the oracle's
    dPhi-based formulas; partial last row tiles, several row chunks, isotropic (dimension-0 quirk) and ARD."""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(N + n)
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    if ard:
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()))
        ls = np.linspace(0.7, 1.5, d)
    else:
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=3)
        ls = 1.2
    var, reg = 0.3, 1.4
    Phi = orc.rff_transform(X, basis.W, ls)
    dP = orc.rff_grad(X, basis.W, ls)
    o = orc.slm_elbo(Phi, y, var, np.full(2 * n, reg), slice(None), [dP[:, :, i] for i in range(d)] if ard else [dP])
    if chunk:
        monkeypatch.setenv("RR_PASS2_CHUNK_ROWS", str(chunk))
    out = []
    for nofuse in ("0", "1"):
        monkeypatch.setenv("RR_PASS2_NO_FUSE", nofuse)
        st = basis.device_fit_state(X, y)
        sq, dh = st.second_pass(ls, o["m"], o["C"], var)
        st.release()
        out.append((sq, np.atleast_1d(np.asarray(dh, dtype=float))))
    err = y - Phi @ o["m"]
    for sq, dh in out:
        assert abs(sq - err @ err) < 1e-4 * (err @ err)
        assert normwise(dh, -np.atleast_1d(np.array(o["dhyp"], dtype=float)).ravel()) < 2e-3
    assert normwise(out[0][1], out[1][1]) < 2e-4


@pytest.mark.parametrize("N,d,n,ard", [(3000, 5, 128, True), (1100, 32, 256, False), (700, 17, 384, True)])
def test_float64_second_pass_product_fused_with_its_contraction(monkeypatch, N, d, n, ard):
    """The float64 pipeline's second pass (dtype="f64" bases) with U = Phi C contracted in registers on the f64 matrix
    cores (rr_gemm_gradt_f64_kernel, 128-column tiles): sqErr and hyper-gradient against the float64 oracle at float64
    tolerances, and against the stored route; partial last row tiles, isotropic and ARD.  (Opt-in, RR_PASS2_FUSE_F64=1: as fast as
    the stored route, not faster -- DESIGN 3.16.)"""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(N + n)
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    if ard:
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()), dtype="f64")
        ls = np.linspace(0.7, 1.5, d)
    else:
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=3, dtype="f64")
        ls = 1.2
    var, reg = 0.3, 1.4
    Phi = orc.rff_transform(X, basis.W, ls)
    dP = orc.rff_grad(X, basis.W, ls)
    o = orc.slm_elbo(Phi, y, var, np.full(2 * n, reg), slice(None), [dP[:, :, i] for i in range(d)] if ard else [dP])
    out = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("RR_PASS2_FUSE_F64", fuse)
        st = basis.device_fit_state(X, y)
        sq, dh = st.second_pass(ls, o["m"], o["C"], var)
        st.release()
        out.append((sq, np.atleast_1d(np.asarray(dh, dtype=float))))
    err = y - Phi @ o["m"]
    want = -np.atleast_1d(np.array(o["dhyp"], dtype=float)).ravel()
    for sq, dh in out:
        assert abs(sq - err @ err) < 1e-10 * (err @ err)
        assert normwise(dh, want) < 1e-8
    assert normwise(out[0][1], out[1][1]) < 1e-10


@pytest.mark.parametrize("n,extra", [(126, 3), (127, 1), (128, 0), (255, 2), (60, 5), (383, 1)])
def test_concat_gram_phi_t_y_rider_and_fallback(n, extra):
    """Phi^T y of a concatenation rides along with the SYRK in the first pad column of the device feature matrix when the
    total width is not a multiple of 256 (F = 2n + extra: one spare column at 255 / 767, many at 123 / 512+), and falls
    back to its own kernel when it is (F = 256, 512); N not a multiple of 32; two calls in a row and a y-less call
    afterwards show the pad column is clean again."""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(n + extra)
    N, d = 2077, 5
    X = rs.randn(N, d)
    y = np.cos(X @ rs.randn(d)) + 0.05 * rs.randn(N)
    ls = np.linspace(0.8, 1.3, d)
    base = bs.RandomRBF(nbases=n, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive()))
    cols = [orc.rff_transform(X, base.W, ls)]
    if extra:
        base = base + bs.LinearBasis(onescol=False, apply_ind=list(range(extra)))
        cols.append(X[:, :extra])
    else:
        base = base + bs.LinearBasis(onescol=False, apply_ind=[0]) + bs.LinearBasis(onescol=False, apply_ind=[1])
        cols += [X[:, :1], X[:, 1:2]]
        base = bs.RandomRBF(nbases=n - 1, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive())) \
            + bs.LinearBasis(onescol=False, apply_ind=[0]) + bs.LinearBasis(onescol=False, apply_ind=[1])
        cols[0] = orc.rff_transform(X, base.bases[0].W, ls)
    ref = np.hstack(cols)
    F = ref.shape[1]
    for rep in range(2):
        G, b, yty = base.gram(X, y, ls)
        assert G.shape == (F, F) and np.array_equal(G, G.T)
        assert normwise(G, ref.T @ ref) < 1e-4 and normwise(b, ref.T @ y) < 1e-4 and abs(yty - y @ y) < 1e-5 * (y @ y)
    G2, b2, _ = base.gram(X, None, ls)
    assert b2 is None and normwise(G2, ref.T @ ref) < 1e-4


def test_concat_gram_on_device_vs_oracle():
    """BASELINE config 3 shape in miniature: RandomMatern52 + LinearBasis(onescol) (+ a generic basis),
    Phi assembled on the device, one SYRK."""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(8)
    N, d, n = 3001, 12, 150
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    ls = np.linspace(0.7, 1.4, d)
    base = bs.RandomMatern52(nbases=n, Xdim=d, random_state=4, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True) + bs.BiasBasis(offset=2.0) \
        + bs.FastFoodRBF(nbases=20, Xdim=2, random_state=5, apply_ind=[3, 1])
    G, b, yty = base.gram(X, y, ls, 0.9)
    Phi = base.transform(X, ls, 0.9)       # per-child GPU transforms, host hstack
    W = base.bases[0].W
    B, Gm, PI, S = orc.fastfood_matrices(20, 2, 5)
    ref = np.hstack((orc.rff_transform(X, W, ls), orc.linear_transform(X), np.full((N, 1), 2.0),
                     orc.fastfood_transform(X[:, [3, 1]], B, Gm, PI, S, 0.9)))
    assert normwise(Phi, ref) < 1e-3
    F = ref.shape[1]
    assert G.shape == (F, F) and np.array_equal(G, G.T)
    assert normwise(G, ref.T @ ref) < 1e-4
    assert normwise(b, ref.T @ y) < 1e-4 and abs(yty - y @ y) < 1e-5 * (y @ y)
    G2, b2, t2 = base.gram(X, None, ls, 0.9)
    assert b2 is None and normwise(G2, ref.T @ ref) < 1e-4


def test_resident_concat_elbo_matches_reference(golden):
    """`_elbo` of a concatenation (config 3's RandomMatern52 + LinearBasis) with X, y resident on the device
    and Phi / dPhi never built: against the reference's golden values."""
    bs, Parameter, Positive, SLM = _imports()
    g = golden("elbo")
    X, y = g["X"], g["y"]
    d, n = X.shape[1], 16
    basis = bs.RandomMatern52(nbases=n, Xdim=d, random_state=22,
                              lenscale=Parameter(np.ones(d), Positive())) + bs.LinearBasis(onescol=True)
    slm = SLM(basis)
    slm.obj_ = -np.inf
    slm._state = basis.device_fit_state(X, y)
    assert slm._state is not None
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, float(g["var"]), list(g["cat_reg"]), g["cat_ls"])
    slm._state.release()
    assert abs(-nelbo - g["cat_elbo"]) < 1e-4 * abs(g["cat_elbo"])
    assert normwise(slm.weights_, g["cat_m"]) < 1e-3
    assert normwise(slm.covariance_, g["cat_C"]) < 1e-3
    assert normwise(-np.asarray(ndreg), g["cat_dreg"]) < 1e-3
    assert np.shape(ndhyp) == (d,) and normwise(-ndhyp, g["cat_dhyp"]) < 2e-3


@pytest.mark.parametrize("chunk_rows,n0,n1,nofuse", [(None, 70, 90, None), (512, 70, 90, None), (None, 256, 512, None),
                                                     (512, 256, 512, None), (None, 256, 512, "1"), (None, 256, 90, None)])
def test_concat_second_pass_and_predict_vs_oracle(monkeypatch, chunk_rows, n0, n1, nofuse):
    """Two random-feature children (isotropic on a column subset, ARD) + Linear + Bias: statistics, sqErr, the
    per-child gradient contraction (structured like apply_grad over BasisCat.grad) and predict_moments.  With both
    children in whole 256-column tiles (n0 = 256, n1 = 512) U = Phi C is contracted child by child in registers and
    never stored (rr_featmat_pass2_rows_planned; RR_PASS2_NO_FUSE=1: the stored route at the same widths; n1 = 90: a
    plan that cannot be fused falls back for BOTH children)."""
    bs, Parameter, Positive, SLM = _imports()
    from revrand_amd.basis_functions import CatFitState
    if nofuse:
        monkeypatch.setenv("RR_PASS2_NO_FUSE", nofuse)
    rs = np.random.RandomState(5)
    N, d = 1300, 6
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    cat = bs.RandomRBF(nbases=n0, Xdim=2, random_state=1, apply_ind=[4, 1]) \
        + bs.RandomMatern32(nbases=n1, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=False) + bs.BiasBasis(offset=1.5)
    ls0, ls1 = 0.8, np.linspace(0.7, 1.5, d)
    var = 0.3
    W0, W1 = cat.bases[0].W, cat.bases[1].W
    Xa = X[:, [4, 1]]
    Phi = np.hstack((orc.rff_transform(Xa, W0, ls0), orc.rff_transform(X, W1, ls1), X, np.full((N, 1), 1.5)))
    F = Phi.shape[1]
    ends = [0, 2 * n0, 2 * n0 + 2 * n1]
    dPs = []
    g0 = np.zeros((N, F))
    g0[:, ends[0]:ends[1]] = orc.rff_grad(Xa, W0, ls0)
    dPs.append(g0)
    g1 = orc.rff_grad(X, W1, ls1)
    for i in range(d):
        gi = np.zeros((N, F))
        gi[:, ends[1]:ends[2]] = g1[:, :, i]
        dPs.append(gi)
    L = np.concatenate((np.full(2 * n0, 1.1), np.full(2 * n1, 0.9), np.full(d, 2.0), [1.3]))
    o = orc.slm_elbo(Phi, y, var, L, slice(None), dPs)

    st = cat.device_fit_state(X, y)
    assert isinstance(st, CatFitState)
    if chunk_rows:
        st.release()
        st = CatFitState(cat, [b._resident_child(X) for b in cat.bases], X, y, chunk_rows=chunk_rows)
    G, b, yty = st.gram([ls0, ls1])
    assert normwise(G, Phi.T @ Phi) < 1e-4 and normwise(b, Phi.T @ y) < 1e-4 and abs(yty - y @ y) < 1e-5 * (y @ y)
    sq, dh = st.second_pass([ls0, ls1], o["m"], o["C"], var)
    st.release()
    err = y - Phi @ o["m"]
    assert abs(sq - err @ err) < 1e-4 * (err @ err)
    assert isinstance(dh, list) and len(dh) == 2 and np.ndim(dh[0]) == 0 and np.shape(dh[1]) == (d,)
    want = -np.array(o["dhyp"])
    assert normwise(np.concatenate(([dh[0]], dh[1])), want) < 2e-3

    Xs = rs.randn(300, d)
    Ey, Vf = cat.predict_moments(Xs, [ls0, ls1], o["m"], o["C"])
    Ps = np.hstack((orc.rff_transform(Xs[:, [4, 1]], W0, ls0), orc.rff_transform(Xs, W1, ls1), Xs, np.full((300, 1), 1.5)))
    Eo, Vo = orc.slm_predict_moments(Ps, o["m"], o["C"], 0.0)
    assert normwise(Ey, Eo) < 1e-4 and normwise(Vf, Vo) < 1e-3


def test_concat_fit_runs_resident_and_predicts():
    """fit() of RandomMatern52 (ARD) + LinearBasis stays on the device-resident path and generalises."""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(0)
    N, d = 4000, 4
    X = rs.randn(N, d)
    f = lambda Z: np.sin(2 * Z[:, 0]) + 0.5 * Z[:, 1] - 0.2 * Z[:, 2]
    y = f(X) + 0.05 * rs.randn(N)
    cat = bs.RandomMatern52(nbases=150, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True)
    slm = SLM(cat, nstarts=0, maxiter=60, random_state=3)
    made = []
    orig = slm._make_state
    slm._make_state = lambda X_, y_: made.append(orig(X_, y_)) or made[-1]
    slm.fit(X, y)
    assert made and type(made[0]).__name__ == "CatFitState"
    Xs = rs.randn(500, d)
    Ey, Vy = slm.predict_moments(Xs)
    assert smse(f(Xs), Ey) < 0.5 and np.all(Vy > 0)   # L-BFGS paths vary in the last bits; the oracle check below is exact
    # device predict_moments of the concatenation == host formulas on the transformed features
    Phi = cat.transform(Xs, *np.atleast_1d([slm.hypers_]) if np.ndim(slm.hypers_) == 0 else [slm.hypers_])
    Eo, Vo = orc.slm_predict_moments(Phi, slm.weights_, slm.covariance_, slm.var_)
    assert normwise(Ey, Eo) < 1e-3
    # phi^T C phi in f32: the error scales with |phi|^T |C| |phi| (cancellation when the optimiser ends at a badly scaled
    # posterior -- its path varies in the last bits with the order of the f64 atomics), not with the result
    C = slm.covariance_
    bound = ((np.abs(Phi) @ np.abs(C)) * np.abs(Phi)).sum(axis=1)
    assert np.all(np.abs(Vy - Vo) <= 1e-3 * Vo + 3e-6 * bound)


@pytest.mark.parametrize("F", [1, 37, 128, 129, 512, 1300])
def test_posterior_on_device_vs_host_solve_posdef(F):
    """rr_posterior_dev: C = (diag(iL) + G/var)^-1, m, log|iC|, sum(G o C), diag(C) against the host
    solve_posdef (mathfun/linalg.py:84-125) -- float64 both sides."""
    from revrand_amd import _hip
    from revrand_amd.linalg import solve_posdef
    assert _hip.posterior_available()
    assert not _hip.posterior_available(64) and _hip.posterior_available(4096)   # default: only from F >= 256
    dev = _hip.get_device()
    rs = np.random.RandomState(F)
    A = rs.randn(F, 3 * F)
    G = A @ A.T
    b = rs.randn(F)
    iL = 1. / rs.gamma(2., 1., F)
    var = 0.37
    acc = dev.upload_vector(np.concatenate((G.ravel(), b)))
    dC = dev.malloc(F * F * 8)
    pG, pb = _hip.ctypes.c_void_p(acc.ptr.value), _hip.ctypes.c_void_p(acc.ptr.value + F * F * 8)
    m, dg, logdet, tr = dev.posterior(F, pG, pb, iL, var, dC)
    C = dev.download(dC, (F, F), np.float64)
    Ch, ldh = solve_posdef(np.diag(iL) + G / var, np.eye(F))
    assert np.array_equal(C, C.T)
    assert normwise(C, Ch) < 1e-9 and abs(logdet - ldh) < 1e-9 * abs(ldh)
    assert normwise(m, Ch @ b / var) < 1e-9 and normwise(dg, Ch.diagonal()) < 1e-9
    assert abs(tr - (G * Ch).sum()) < 1e-9 * abs((G * Ch).sum())
    # a matrix whose Cholesky is not safe (diag below CHOLTHRESH = 1e-5): the call reports it, nothing raised
    Gs = np.zeros((F, F))
    acc2 = dev.upload_vector(np.concatenate((Gs.ravel(), b)))
    p2 = _hip.ctypes.c_void_p(acc2.ptr.value)
    assert dev.posterior(F, p2, _hip.ctypes.c_void_p(acc2.ptr.value + F * F * 8), np.full(F, 1e-12), 1.0, dC) is None
    for buf in (acc, acc2, dC):
        buf.free()


def test_elbo_device_posterior_equals_host_posterior(monkeypatch):
    """One resident `_elbo` evaluation with the posterior on the device and (RR_POSDEF=host) on the host: objective,
    gradients, weights and covariance agree (float64 linear algebra on both sides); the fit then fetches the best
    covariance from the device once."""
    bs, Parameter, Positive, SLM = _imports()
    from revrand_amd import _hip
    X, y, Xs, ys = _gaus_data()
    d = X.shape[1]
    res = []
    for mode in ("device", "host"):
        monkeypatch.setenv("RR_POSDEF", mode)
        assert _hip.posterior_available(80) == (mode == "device")
        cat = bs.RandomRBF(nbases=40, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive())) \
            + bs.LinearBasis(onescol=True)
        for basis, reg, hyp in ((cat.bases[0], 1.3, np.full(d, 0.8)), (cat, [1.3, 0.7], np.full(d, 0.8))):
            slm = SLM(basis)
            slm.obj_ = -np.inf
            slm._state = basis.device_fit_state(X, y)
            f, (gv, gr, gh) = slm._elbo(X, y, 0.4, reg, hyp)
            slm._state.release()
            res.append((f, gv, np.atleast_1d(gr), np.atleast_1d(gh), slm.weights_, slm.covariance_))
    for dev_r, host_r in ((res[0], res[2]), (res[1], res[3])):
        # (the two evaluations recompute the f32 statistics, whose atomically accumulated last bits differ run to run;
        # the float64 algebra itself agrees to 1e-9: test_posterior_on_device_vs_host_solve_posdef)
        assert abs(dev_r[0] - host_r[0]) < 1e-6 * abs(host_r[0]) and abs(dev_r[1] - host_r[1]) < 1e-5 * abs(host_r[1])
        assert normwise(dev_r[2], host_r[2]) < 1e-5 and normwise(dev_r[3], host_r[3]) < 1e-3
        assert normwise(dev_r[4], host_r[4]) < 1e-5 and normwise(dev_r[5], host_r[5]) < 1e-5
    monkeypatch.setenv("RR_POSDEF", "device")
    basis = bs.RandomRBF(nbases=40, Xdim=d, random_state=3)
    slm = SLM(basis, nstarts=0, maxiter=40, random_state=0)
    best, orig = [], SLM._elbo_resident

    def recording(self, X_, y_, var, reg, hypers):   # parameters of the evaluation that set the best posterior
        before = self.obj_
        out = orig(self, X_, y_, var, reg, hypers)
        if self.obj_ > before:
            best[:] = [var, reg, hypers]
        return out

    monkeypatch.setattr(SLM, "_elbo_resident", recording)
    slm.fit(X, y)
    assert slm.covariance_.shape == (80, 80) and normwise(slm.covariance_, slm.covariance_.T) < 1e-6
    assert np.all(np.isfinite(slm.predict(Xs)))   # (fit quality varies with the L-BFGS path; the pairing below is exact)
    # the covariance fetched at the end of fit belongs to the best evaluation, like weights_
    var, reg, hyp = best
    G, b, _ = basis.gram(X, y, hyp)
    C = np.linalg.inv(np.eye(80) / reg + G / var)
    assert normwise(slm.covariance_, C) < 1e-4 and normwise(slm.weights_, C @ b / var) < 1e-4


def test_config3_width_concat_gram_properties():
    """BASELINE config 3's shape at one GPU's share in miniature rows (RandomMatern52 n=4096 + LinearBasis(onescol),
    D=64 -> F_tot = 8257, 30k rows): the device-assembled Gram through identities that need no CPU oracle --
    symmetry, cos^2 + sin^2 per frequency, the exact [1, X] block, and shard linearity of the resident fit state."""
    bs, Parameter, Positive, SLM = _imports()
    N, d, n = 30_000, 64, 4096
    rs = np.random.RandomState(2)
    X = rs.randn(N, d).astype(np.float32).astype(np.float64)
    y = np.sin(X[:, 0]) + 0.1 * rs.randn(N)
    cat = bs.RandomMatern52(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True)
    ls = np.linspace(0.8, 1.6, d)
    st = cat.device_fit_state(X, y)
    G, b, yty = st.gram([ls])
    F = 2 * n + d + 1
    assert G.shape == (F, F) and np.array_equal(G, G.T)
    dg = np.diag(G)
    assert np.abs(dg[:n] + dg[n:2 * n] - N / n).max() < 2e-5 * (N / n)
    lin = np.hstack((np.ones((N, 1)), X))
    assert normwise(G[2 * n:, 2 * n:], lin.T @ lin) < 1e-5 and normwise(b[2 * n:], lin.T @ y) < 1e-5
    assert abs(yty - y @ y) < 1e-6 * (y @ y)
    st.release()
    # shard linearity: two halves through the same code path sum to the whole
    parts = []
    for sl in (slice(0, N // 2), slice(N // 2, N)):
        sts = cat.device_fit_state(X[sl], y[sl])
        parts.append(sts.gram([ls]))
        sts.release()
    assert normwise(parts[0][0] + parts[1][0], G) < 1e-5 and normwise(parts[0][1] + parts[1][1], b) < 1e-5


@pytest.mark.parametrize("tag", ["iso", "ard"])
def test_resident_elbo_f64_matches_reference(golden, tag):
    """dtype='f64' with (X, y) resident: f64 features, f64 MFMA Gram, f64 posterior, f64 second pass (f64 MFMA
    GEMM): every output of `_elbo`, the hyper-gradient included, within the BASELINE fp64 tolerance of the
    reference's golden values; and predict_moments in f64."""
    bs, Parameter, Positive, SLM = _imports()
    g = golden("elbo")
    X, y = g["X"], g["y"]
    d, n = X.shape[1], 16
    lsp = Parameter(1., Positive()) if tag == "iso" else Parameter(np.ones(d), Positive())
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=21, lenscale=lsp, dtype="f64")
    slm = SLM(basis)
    slm.obj_ = -np.inf
    slm._state = basis.device_fit_state(X, y)
    assert slm._state is not None
    ls = float(g[tag + "_ls"]) if tag == "iso" else g[tag + "_ls"]
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, float(g["var"]), float(g["reg"]), ls)
    slm._state.release()
    assert abs(-nelbo - g[tag + "_elbo"]) < 1e-9 * abs(g[tag + "_elbo"])
    assert normwise(slm.weights_, g[tag + "_m"]) < 1e-8 and normwise(slm.covariance_, g[tag + "_C"]) < 1e-8
    assert normwise(-ndvar, g[tag + "_dvar"]) < 1e-8
    assert normwise(-np.atleast_1d(ndreg), g[tag + "_dreg"]) < 1e-8
    assert normwise(-np.atleast_1d(ndhyp), g[tag + "_dhyp"]) < 1e-7
    Xs = np.random.RandomState(3).randn(700, d)
    Ey, Vf = basis.predict_moments(Xs, ls, slm.weights_, slm.covariance_)
    Eo, Vo = orc.slm_predict_moments(orc.rff_transform(Xs, basis.W, ls), slm.weights_, slm.covariance_, 0.0)
    assert normwise(Ey, Eo) < 1e-10 and normwise(Vf, Vo) < 1e-10


def test_second_pass_f64_ragged_multi_chunk(monkeypatch):
    """f64 second pass over several row chunks with ragged sizes against the oracle's dPhi-based formulas."""
    bs, Parameter, Positive, SLM = _imports()
    monkeypatch.setenv("RR_PASS2_CHUNK_ROWS", "512")
    N, d, n = 1333, 7, 75
    rs = np.random.RandomState(12)
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    basis = bs.RandomMatern32(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()), dtype="f64")
    ls = np.linspace(0.7, 1.5, d)
    Phi = orc.rff_transform(X, basis.W, ls)
    dP = orc.rff_grad(X, basis.W, ls)
    o = orc.slm_elbo(Phi, y, 0.3, np.full(2 * n, 1.4), slice(None), [dP[:, :, i] for i in range(d)])
    st = basis.device_fit_state(X, y)
    G, b, yty = st.gram(ls)
    sq, dh = st.second_pass(ls, o["m"], o["C"], 0.3)
    st.release()
    err = y - Phi @ o["m"]
    assert normwise(G, Phi.T @ Phi) < 1e-10 and abs(sq - err @ err) < 1e-10 * (err @ err)
    assert normwise(dh, -np.array(o["dhyp"])) < 1e-8


def test_gridsearch_in_worker_processes_like_reference_test_models():
    """tests/test_models.py:53-80,166-194 of the reference: GridSearchCV / RandomizedSearchCV over estimators, in
    worker PROCESSES (n_jobs=2): estimators and bases are pickled to the workers (device handles are per process,
    created lazily, never pickled) and each worker drives the GPU through its own context."""
    from sklearn.model_selection import GridSearchCV, RandomizedSearchCV
    bs, Parameter, Positive, SLM = _imports()
    from revrand_amd.glm import GeneralizedLinearModel
    from revrand_amd.likelihoods import Gaussian
    X, y, Xs, ys = _gaus_data()
    slm = SLM(bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=20, Xdim=X.shape[1], random_state=0), nstarts=0,
              maxiter=30)
    slm.basis.transform(X[:5], 1.0)   # the parent holds live handles when the estimator is pickled
    est = GridSearchCV(slm, {"var": [Parameter(v, Positive()) for v in [1.0, 2.0]]}, n_jobs=2, cv=3)
    est.fit(X, y)
    assert len(est.predict(Xs)) == len(ys) and smse(ys, est.predict(Xs)) < 0.5
    est = RandomizedSearchCV(SLM(bs.LinearBasis(onescol=True)), {"var": [Parameter(1.0 / v, Positive()) for v in range(1, 6)]},
                             n_jobs=2, n_iter=2, cv=3, random_state=0)
    est.fit(X, y)
    assert len(est.predict(Xs)) == len(ys)
    glm = GeneralizedLinearModel(Gaussian(), bs.LinearBasis(onescol=True), random_state=1, maxiter=100, nstarts=10)
    est = GridSearchCV(glm, {"batch_size": [10, 20]}, n_jobs=2, cv=3)
    est.fit(X, y)
    assert len(est.predict(Xs)) == len(ys)
    # joblib keeps its worker processes for reuse, each with a device context on the one GPU: ended with the test
    from joblib.externals.loky import get_reusable_executor
    get_reusable_executor().shutdown(wait=True, kill_workers=True)


def _gloo_gpu_fit_worker(rank, world, port, q, n):
    """One rank of a row-sharded fit with the REAL device state (both ranks share GPU 0, each with its own context);
    gloo carries the two all-reduces of every `_elbo` on the host."""
    import os
    import torch.distributed as dist
    from revrand_amd import parallel
    bs, Parameter, Positive, SLM = _imports()
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(3)
        N, d = 6001, 3
        X = rs.randn(N, d)
        y = np.sin(X @ np.array([1.0, -0.5, 0.3])) + 0.3 * rs.randn(N)
        a, b = parallel.shard_bounds(N, rank, world)
        from scipy.stats import gamma
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=5, lenscale=Parameter(np.ones(d), Positive()),
                             regularizer=Parameter(gamma(1.), Positive()))
        # a random regulariser + nstarts: every rank ranks the same three starts (same seed) by the objective-only
        # evaluation of the summed statistics
        slm = SLM(basis, var=Parameter(0.1, Positive()), nstarts=3, maxiter=8, distributed=world > 1, random_state=0)
        slm.obj_ = -np.inf
        slm._state = slm._make_state(X[a:b], y[a:b])
        assert slm._state is not None
        ls = np.array([0.8, 1.1, 1.4])
        f, (g_var, g_reg, g_hyp) = slm._elbo(X[a:b], y[a:b], 0.2, 1.3, ls)
        ev = [float(f), float(g_var), float(g_reg)] + np.asarray(g_hyp).tolist() + slm.weights_[:8].tolist()
        slm._state.release()
        slm._state = None
        slm.fit(X[a:b], y[a:b])
        Ey = slm.predict(X[:50])
        q.put((rank, ev, float(slm.var_), float(slm.regularizer_), np.asarray(slm.hypers_).tolist(), float(slm.obj_),
               Ey.tolist()))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n", [12, 160])
def test_two_rank_gloo_fit_with_real_device_state(n):
    """The distributed estimator with real kernels on both ranks (tests/test_dist_gloo.py runs the same protocol with a
    NumPy stand-in on CPU): the ranks end bit-identical, and one sharded `_elbo` equals the single-process one to f32
    accuracy.  n = 160 (F = 320) is above the device-posterior threshold, which a gloo group sends down the host route."""
    import multiprocessing as pymp
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_gpu_fit_worker, args=(r, 2, port, q, n)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=600) for _ in procs)
    finally:  # a worker that has not exited 120 s after its result is ended and fails the test (never joined forever at exit)
        codes = []
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
                p.join(10)
            codes.append(p.exitcode)
        assert codes == [0, 0], codes
    q1 = pymp.Queue()
    _gloo_gpu_fit_worker(0, 1, 0, q1, n)
    single = q1.get(timeout=10)
    assert res[0][1:] == res[1][1:]                       # identical statistics -> identical L-BFGS path on both ranks
    assert np.allclose(res[0][1], single[1], rtol=2e-4, atol=2e-4 * np.abs(single[1]).max())
    assert np.isfinite(res[0][5]) and np.isfinite(res[0][6]).all()


@pytest.mark.parametrize("n", [20, 160])
def test_objective_only_evaluation_equals_full_elbo(n):
    """`_elbo_objective` (statistics pass + posterior, no second data pass: sqErr from y^T y, m^T b and the posterior
    identity) gives the objective of the full `_elbo`, for the host posterior (F = 40) and the device one (F = 320), a
    single basis and a concatenation; `fit` ranks its random starts with it and only then runs full evaluations."""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(1)
    N, d = 5000, 4
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.2 * rs.randn(N)
    for basis, hyp, reg in (
            (bs.RandomRBF(nbases=n, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive())),
             np.linspace(0.7, 1.4, d), 1.3),
            (bs.RandomMatern32(nbases=n, Xdim=d, random_state=3) + bs.LinearBasis(onescol=True), [0.9], [1.1, 2.0])):
        slm = SLM(basis)
        slm.obj_ = -np.inf
        slm._state = slm._make_state(X, y)
        assert slm._state is not None
        for var in (0.05, 0.7):
            full = slm._elbo(X, y, var, reg, hyp)[0]
            # sqErr = y^T y - m^T b - ... cancels ~1 digit per factor y^T y / sqErr of f32 statistics: far below what
            # separates random starts
            assert abs(slm._elbo_objective(X, y, var, reg, hyp) - full) < 5e-5 * abs(full)
        slm._state.release()
        slm._state = None
    # fit: the random starts do not run the second pass
    from revrand_amd.basis_functions import DeviceFitState
    calls = {"second": 0, "gram": 0}
    o2, og = DeviceFitState.second_pass, DeviceFitState.gram_device
    og_host = DeviceFitState.gram

    def spy2(self, *a, **k):
        calls["second"] += 1
        return o2(self, *a, **k)

    def spyg(self, *a, **k):
        calls["gram"] += 1
        return og(self, *a, **k)

    def spygh(self, *a, **k):
        calls["gram"] += 1
        return og_host(self, *a, **k)
    DeviceFitState.second_pass, DeviceFitState.gram_device, DeviceFitState.gram = spy2, spyg, spygh
    try:
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=2)
        slm = SLM(basis, nstarts=12, maxiter=5, random_state=4).fit(X, y)
    finally:
        DeviceFitState.second_pass, DeviceFitState.gram_device, DeviceFitState.gram = o2, og, og_host
    assert calls["gram"] >= calls["second"] + 12 and calls["second"] >= 1
    assert np.isfinite(slm.predict(X[:10])).all()


def test_predict_is_the_mean_of_predict_moments_without_the_variance_product():
    """`predict` forms Phi m alone on the device (the reference goes through `predict_moments`, slm.py:201-217): same
    numbers as the mean of `predict_moments`, for a single basis, a concatenation with generic children and a purely
    linear model; an f64 basis keeps the `predict_moments` route."""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(0)
    X = rs.randn(3000, 5)
    y = np.sin(X[:, 0]) + 0.3 * X[:, 1] + 0.1 * rs.randn(3000)
    Xs = rs.randn(777, 5)
    for basis in (bs.RandomRBF(nbases=100, Xdim=5, random_state=1),
                  bs.RandomMatern32(nbases=60, Xdim=2, random_state=2, apply_ind=[0, 1]) + bs.LinearBasis(onescol=True)
                  + bs.FastFoodRBF(nbases=8, Xdim=2, random_state=3, apply_ind=[2, 3]),
                  bs.LinearBasis(onescol=True),
                  bs.RandomRBF(nbases=40, Xdim=5, random_state=1, dtype="f64")):
        slm = SLM(basis, maxiter=15, nstarts=0, random_state=0).fit(X, y)
        Em, _ = slm.predict_moments(Xs)
        Ey = slm.predict(Xs)
        assert Ey.shape == (777,) and normwise(Ey, Em) < 1e-5
        assert ((slm.predict(X) - y) ** 2).mean() < 0.9 * y.var()


def test_predict_of_a_random_kernel_basis_comes_from_the_feature_kernel_alone(monkeypatch):
    """`predict` (slm.py:201-217) of a single f32 random kernel basis: rr_rff_predict_mean_dev -- the very dot products
    `predict_moments` returns as Ey (same kernel, same bits), with no covariance on the device and no feature matrix; ragged
    row counts, f32 / f64 queries, `apply_ind`, ARD; validation of a large query on the second host thread with sklearn's
    exception for a non-finite row; float64-phase and Xdim > 128 bases keep the GEMM route."""
    import threading
    from revrand_amd import _hip
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(5)
    calls = []
    orig = _hip.RffHandle._predict
    monkeypatch.setattr(_hip.RffHandle, "_predict",
                        lambda self, X, N, lsp, nls, m, C, f: (calls.append((C is None, [t.name for t in threading.enumerate()])),
                                                                orig(self, X, N, lsp, nls, m, C, f))[1])
    for d, n, ard, ai, xdt in ((5, 100, False, None, np.float64), (9, 129, True, None, np.float32), (3, 64, False, [4, 0, 2], np.float64)):
        X = rs.randn(4000, 5 if ai else d).astype(xdt)
        y = (np.sin(X[:, 0]) + 0.1 * rs.randn(4000)).astype(xdt)
        ls = Parameter(np.full(d, 1.3), Positive()) if ard else Parameter(1.3, Positive())
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=ls, apply_ind=ai)
        slm = SLM(basis, maxiter=8, nstarts=0, random_state=0).fit(X, y)
        for N in (1, 255, 257, 4000):
            del calls[:]
            Ey = slm.predict(X[:N])
            assert calls == [(True, calls[0][1])] and "_serve" not in slm.__dict__   # nothing uploaded for it
            Em, _ = slm.predict_moments(X[:N])
            assert Ey.shape == (N,) and np.array_equal(Ey, Em)
            slm._drop_serving()
        Xa = X[:, ai] if ai else X
        Phi = orc.rff_transform(Xa[:500].astype(np.float64), basis.W, slm.hypers_)
        assert normwise(slm.predict(X[:500]), Phi @ slm.weights_) < 1e-4
    # a large query: validated while it is uploaded
    # (the validation of 40 000 x 5 values can be over before the wrapper above looks at the running threads: what is
    # recorded is that the thread was MADE)
    XL = np.tile(X, (10, 1))
    del calls[:]
    made, plain_thread = [], threading.Thread

    class RecordingThread(plain_thread):
        def __init__(self, *a, **k):
            made.append(k.get("name"))
            super().__init__(*a, **k)
    monkeypatch.setattr(_hip.threading, "Thread", RecordingThread)
    EL = slm.predict(XL)
    monkeypatch.setattr(_hip.threading, "Thread", plain_thread)
    assert calls[0][0] and "rr-predict-check" in made and np.array_equal(EL[:4000], slm.predict(X))
    for bad_row in (0, len(XL) - 1):
        Xb = XL.copy()
        Xb[bad_row, 2] = np.nan
        with pytest.raises(ValueError, match="NaN"):
            slm.predict(Xb)
        with pytest.raises(ValueError, match="NaN"):
            slm.predict(Xb[bad_row:][:1] if bad_row == 0 else Xb[-3:])
    with pytest.raises((ValueError, IndexError)):
        slm.predict(XL[:, :2])          # wrong width: the basis' own check (apply_ind's IndexError here, as in the reference)
    # bases without the route: the 256-column product, same numbers as predict_moments' mean
    for basis in (bs.RandomLaplace(nbases=64, Xdim=5, random_state=2), bs.RandomRBF(nbases=32, Xdim=130, random_state=3)):
        Xw = rs.randn(1500, basis.d)
        slm = SLM(basis, maxiter=5, nstarts=0, random_state=0).fit(Xw, np.sin(Xw[:, 0]))
        del calls[:]
        Ey = slm.predict(Xw)
        assert not any(c[0] for c in calls) and slm._serve["feats"] is not None
        assert normwise(Ey, slm.predict_moments(Xw)[0]) < 1e-5


def test_serving_state_is_reused_and_invalidated():
    """A fitted estimator keeps its covariance in HBM and its `predict` feature matrix between calls; the state is
    rebuilt when `covariance_` is replaced, dropped by pickling and by a new fit."""
    import pickle
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(0)
    X = rs.randn(2000, 4)
    y = np.sin(X[:, 0]) + 0.1 * rs.randn(2000)
    slm = SLM(bs.RandomRBF(nbases=150, Xdim=4, random_state=1), nstarts=0, maxiter=10, random_state=0).fit(X, y)
    Xs = rs.randn(50, 4)
    Ey, Vy = slm.predict_moments(Xs)
    srv = slm._serve
    assert srv["cov"] is not None
    Ey2, Vy2 = slm.predict_moments(Xs)
    assert slm._serve is srv and np.array_equal(Ey, Ey2) and np.array_equal(Vy, Vy2)
    assert np.array_equal(slm.predict(Xs), Ey) and slm._serve["feats"] is None   # the feature kernel alone: no feature matrix
    Phi = orc.rff_transform(Xs, slm.basis.W, slm.hypers_)
    assert normwise(Vy, (Phi @ slm.covariance_ * Phi).sum(axis=1) + slm.var_) < 1e-3
    slm.covariance_ = 4.0 * slm.covariance_                    # replaced: the device copy must follow
    _, Vy4 = slm.predict_moments(Xs)
    assert slm._serve is not srv and normwise(Vy4 - slm.var_, 4.0 * (Vy - slm.var_)) < 1e-5
    clone_ = pickle.loads(pickle.dumps(slm))
    assert "_serve" not in clone_.__dict__ and normwise(clone_.predict_moments(Xs)[1], Vy4) < 1e-6
    slm.fit(X, y)
    assert "_serve" not in slm.__dict__


def test_predict_moments_validates_a_large_query_while_the_gpu_works_on_it(monkeypatch):
    """predict_moments (slm.py:219-244) of a query above RffHandle.PREDICT_CONCURRENT_CHECK_ROWS: sklearn's check_array runs on
    a host thread while the rows are uploaded and the GPU works on them -- the same numbers as validating first, the same
    exception for a non-finite row, raised before anything is returned; small queries are validated up front."""
    import threading
    import revrand_amd.basis_functions as bs
    from revrand_amd import _hip
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    rs = np.random.RandomState(12)
    d, n, N = 7, 96, 150_001
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X[:, 0]) + 0.1 * rs.randn(N)).astype(np.float32)
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(1.1, Positive()))
    slm = StandardLinearModel(basis, var=Parameter(0.3, Positive()), nstarts=0, maxiter=2).fit(X[:5000], y[:5000])
    # (which threads the call MAKES is recorded, not which are running when the upload starts: a fast host can be through
    # with the validation by then)
    made, plain_thread = [], threading.Thread

    class RecordingThread(plain_thread):
        def __init__(self, *a, **k):
            made.append(k.get("name"))
            super().__init__(*a, **k)
    monkeypatch.setattr(_hip.threading, "Thread", RecordingThread)
    Ey, Vy = slm.predict_moments(X)
    assert made.count("rr-predict-check") == 1
    monkeypatch.setattr(_hip.RffHandle, "PREDICT_CONCURRENT_CHECK_ROWS", 10 ** 9)   # validate first
    Ey1, Vy1 = slm.predict_moments(X)
    assert made.count("rr-predict-check") == 1 and np.array_equal(Ey, Ey1) and np.array_equal(Vy, Vy1)
    Phi = orc.rff_transform(X[-300:].astype(np.float64), basis.W, slm.hypers_)
    Eo, Vo = orc.slm_predict_moments(Phi, slm.weights_, slm.covariance_, slm.var_)
    assert normwise(Ey[-300:], Eo) < 1e-3 and normwise(Vy[-300:], Vo) < 1e-3
    monkeypatch.undo()
    for bad_row in (5, N - 1):
        Xb = X.copy()
        Xb[bad_row, 2] = np.inf
        with pytest.raises(ValueError, match="infinity"):
            slm.predict_moments(Xb)
        with pytest.raises(ValueError, match="infinity"):
            slm.predict_moments(Xb[bad_row - 5 if bad_row > 5 else 0:][:64])   # a small query: validated up front
    Ey2, _ = slm.predict_moments(X)               # nothing is left behind by the failed calls
    assert np.array_equal(Ey2, Ey)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("tag,ns", [("ns6", 6), ("ns1", 1)])
def test_fit_random_starts_vs_reference(golden, tag, ns, dtype):
    """`StandardLinearModel.fit` with distributions as initial values (slm.py:112-125; structured_minimizer's random starts,
    decorators.py:79-97, 541-583): the draw order on `random_` (one start, then nstarts candidates of (var, regulariser,
    length scales)), every candidate's -ELBO, the candidate handed to L-BFGS-B (maxiter = 0 returns it) and the stream's end
    state -- against what the reference's own `fit` did (tests/golden/slm_starts.npz)."""
    from scipy.stats import gamma
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    g = golden("slm_starts")
    X, y = g["X"], g["y"]
    d = X.shape[1]
    basis = bs.RandomRBF(nbases=g["W"].shape[1], Xdim=d, random_state=9, lenscale=Parameter(gamma(2., scale=0.5), Positive(), shape=(d,)),
                         regularizer=Parameter(gamma(1.), Positive()), dtype=dtype)
    assert np.array_equal(basis.W, g["W"])
    slm = StandardLinearModel(basis, var=Parameter(gamma(1.), Positive()), nstarts=ns, maxiter=0, random_state=13)
    objs = []
    real = StandardLinearModel._elbo_objective

    def spy(self, *a, **k):
        v = real(self, *a, **k)
        objs.append(v)
        return v
    StandardLinearModel._elbo_objective = spy
    try:
        slm.fit(X, y)
    finally:
        StandardLinearModel._elbo_objective = real
    tol = 1e-5 if dtype == "f32" else 1e-10
    assert len(objs) == ns and normwise(np.array(objs), g[tag + "_cand_objs"]) < tol
    assert abs(slm.var_ - float(g[tag + "_var"])) < 1e-12 * float(g[tag + "_var"])
    assert abs(slm.regularizer_ - float(g[tag + "_reg"])) < 1e-12 * float(g[tag + "_reg"])
    assert normwise(np.asarray(slm.hypers_), g[tag + "_hyp"]) < 1e-12
    assert slm.random_.randn() == float(g[tag + "_end"])
