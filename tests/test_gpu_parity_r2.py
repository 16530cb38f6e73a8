"""
Round-2 parity additions on the GPU:
* BasisCat / apply_ind against the REFERENCE's own outputs (tests/golden/concat.npz; basis_functions.py:1599-1748 and
  the apply_ind case of the reference's tests/test_bases.py:223-238);
* StandardLinearModel.fit at BASELINE config 1's real shape against the reference's fit (tests/golden/fit_c1.npz);
* the f32 hyper-gradient of the second data pass at config 2's size against the f64 path (both on the GPU).
"""
import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise
from test_oracle_golden import c1_data

pytestmark = pytest.mark.gpu


def smse(y_true, y_pred):
    return ((y_true - y_pred) ** 2).sum() / (len(y_true) * y_true.var())


def _imports():
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    return bs, Parameter, Positive, StandardLinearModel


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_concat_transform_grad_regulariser_vs_reference(golden, dtype):
    bs, Parameter, Positive, _ = _imports()
    g = golden("concat")
    X, ls = g["X"], g["ls"]
    N, d = X.shape
    n = 10
    base = bs.RandomMatern52(nbases=n, Xdim=d, random_state=4, lenscale=Parameter(np.ones(d), Positive()), dtype=dtype) \
        + bs.LinearBasis(onescol=True)
    assert np.array_equal(base.bases[0].W, g["W"])
    tol = 1e-3 if dtype == "f32" else 1e-5
    P = base.transform(X, ls)
    assert P.dtype == np.float64 and normwise(P, g["Phi"]) < tol
    assert np.array_equal(P[:, 2 * n:], g["Phi"][:, 2 * n:])          # [1, X]: exact
    grads = list(base.grad(X, ls))
    assert len(grads) == 1 and grads[0].shape == g["dPhi"].shape
    assert normwise(grads[0], g["dPhi"]) < tol
    assert np.all(grads[0][:, 2 * n:, :] == 0)                         # zero padding to the full width
    diag, slices = base.regularizer_diagonal(X, 2.5, 0.5)
    assert np.array_equal(diag, g["regdiag"])
    assert [[s.start, s.stop] for s in slices] == g["slices"].tolist()
    assert base.get_dim(X) == int(g["get_dim"])


def test_concat_apply_ind_vs_reference(golden):
    bs, Parameter, Positive, _ = _imports()
    g = golden("concat")
    X = g["ai_X"]
    base = bs.LinearBasis(onescol=False, apply_ind=[0]) \
        + bs.RandomRBF(Xdim=1, nbases=1, apply_ind=[1], random_state=8) \
        + bs.RandomRBF(Xdim=2, nbases=3, random_state=9, lenscale=Parameter(np.ones(2), Positive()), apply_ind=[1, 0])
    assert np.array_equal(base.bases[1].W, g["ai_W1"]) and np.array_equal(base.bases[2].W, g["ai_W2"])
    ls2 = np.array([0.8, 1.7])
    P = base.transform(X, 1.5, ls2)
    assert normwise(P, g["ai_Phi"]) < 1e-3 and np.array_equal(P[:, 0], X[:, 0])
    g0, g1 = list(base.grad(X, 1.5, ls2))
    assert g0.shape == (20, 9) and g1.shape == (20, 9, 2)
    assert normwise(g0, g["ai_dPhi0"]) < 1e-3 and normwise(g1, g["ai_dPhi1"]) < 1e-3
    # the same concatenation through the device feature matrix (what _elbo uses): Gram == Phi^T Phi of the reference
    y = np.sin(X[:, 0])
    G, b, yty = base.gram(X, y, 1.5, ls2)
    Pr = g["ai_Phi"]
    assert normwise(G, Pr.T @ Pr) < 1e-3 and normwise(b, Pr.T @ y) < 1e-3 and abs(yty - y @ y) < 1e-5 * (y @ y)


def test_fit_config1_shape_vs_reference(golden):
    """BASELINE configs[0]: RandomRBF nbases=256, D=8, N=10k through fit (slm.py:74-140), the reference's own start
    values, nstarts=0, maxiter=20.  L-BFGS trajectories are sensitive: compared at prediction / objective
    level; ONE evaluation at the reference's fitted point is compared tightly."""
    bs, Parameter, Positive, SLM = _imports()
    g = golden("fit_c1")
    X, y, Xs = c1_data()
    basis = bs.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(2.0, Positive()),
                         regularizer=Parameter(10.0, Positive()))
    assert np.array_equal(basis.W[:, :8], g["c1_W_head"])
    # (1) ONE `_elbo` at the reference's fitted point, both data passes on the GPU, against the reference's own
    # evaluation there: objective, every gradient, posterior weights and variances
    for dtype, tol in (("f32", 1e-3), ("f64", 1e-7)):
        bt = bs.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(2.0, Positive()),
                          regularizer=Parameter(10.0, Positive()), dtype=dtype)
        one = SLM(bt)
        one.obj_ = -np.inf
        one._state = one._make_state(X, y)
        try:
            nelbo, (ndvar, ndreg, ndhyp) = one._elbo(X, y, float(g["c1_var_"]), float(g["c1_reg_"]), float(g["c1_hyp_"]))
        finally:
            one._state.release()
            one._state = None
        assert abs(-nelbo - float(g["c1_at_elbo"])) < max(tol * 1e-1, 1e-9) * abs(float(g["c1_at_elbo"]))
        assert normwise(one.weights_, g["c1_at_m"]) < tol
        assert normwise(np.asarray(one.covariance_).diagonal(), g["c1_at_Cdiag"]) < tol
        assert abs(-ndvar - float(g["c1_at_dvar"])) < tol * abs(float(g["c1_at_dvar"]))
        assert normwise(-np.atleast_1d(ndreg), g["c1_at_dreg"]) < tol
        assert normwise(-np.atleast_1d(ndhyp), g["c1_at_dhyp"]) < 2 * tol
    # (2) the fit itself.  In float64 arithmetic the optimiser walks the reference's own trajectory: same end point, same
    # objective, same predictions.
    b64 = bs.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(2.0, Positive()),
                       regularizer=Parameter(10.0, Positive()), dtype="f64")
    s64 = SLM(b64, var=Parameter(0.02, Positive()), nstarts=0, maxiter=20, random_state=0).fit(X, y)
    Ey64, Vy64 = s64.predict_moments(Xs)
    assert abs(s64.obj_ - float(g["c1_obj"])) < 1e-5 * abs(float(g["c1_obj"]))
    assert abs(s64.var_ - float(g["c1_var_"])) < 1e-6 * float(g["c1_var_"])
    assert abs(float(np.atleast_1d(s64.hypers_)[0]) - float(g["c1_hyp_"])) < 1e-6 * float(g["c1_hyp_"])
    assert smse(g["c1_Ey"], Ey64) < 1e-5 and np.all(Vy64 > 0)
    # In float32 the trajectory is the same to 7 digits up to the reference's end point -- where L-BFGS-B's line search
    # stalls: with the ISOTROPIC length scale the reference's own gradient is the dimension-0 slab only
    # (basis_functions.py:866-901), its run ends "ABNORMAL" after one or two iterations from every start (oracle/make_golden.py
    # gen_fit_converged) and evaluates that point seven more times; whether a later trial step escapes depends on the last
    # bits of the objective.  What holds wherever it ends: an objective -- the quantity being optimised -- at least as good
    # as the reference's, and a model, not noise, on the 64 held-out points (SMSE < 0.6; where the float32 run walks on to
    # a HIGHER objective than the reference's end point it measured 0.44 against the reference's 0.19 there: the ELBO is
    # a training objective).  Held-out quality against a reference fit that CONVERGED -- within 5 % of its SMSE -- is what
    # test_fit_converges_to_the_references_optimum below asserts for the float32 estimator.
    slm = SLM(basis, var=Parameter(0.02, Positive()), nstarts=0, maxiter=20, random_state=0).fit(X, y)
    Ey, Vy = slm.predict_moments(Xs)
    assert slm.obj_ > float(g["c1_obj"]) - 1e-6 * abs(float(g["c1_obj"]))
    assert np.all(Vy > 0)
    assert smse(g["c1_ys_true"], Ey) < 0.6


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-5), ("f32", 1e-3)])
def test_fit_converges_to_the_references_optimum(golden, dtype, tol):
    """A CONVERGED reference fit (tests/golden/fit_converged.npz: config 1's shape with ARD length scales, L-BFGS-B
    `success` asserted while generating, 26 iterations) against `fit` from the same start values (slm.py:74-140).

    * What does not depend on where an optimiser stops: ONE `_elbo` at the reference's optimum reproduces its objective
      (1e-9 in float64 arithmetic, 1e-5 in the default float32) and posterior weights (north star's 1e-5 / 1e-3), and the
      point is stationary for this implementation too (log-space gradient below 1e-4 of the objective).
    * The fit itself: objective at `tol` (1e-5 / 1e-3), held-out SMSE within 5 % of the reference's, and the parameters at
      max(1e-4, tol) -- L-BFGS-B stops on a RELATIVE DECREASE of the objective (1e-8), which pins the objective, not the
      argument: last-bit differences of the statistics (floating-point atomics) move the stopping iterate by ~1e-5 from run
      to run, and the reference's own optimum moves by up to 1e-4 between two start points (oracle/make_golden.py)."""
    bs, Parameter, Positive, SLM = _imports()
    g = golden("fit_converged")
    X, y, Xs = c1_data()

    def make():
        return bs.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(np.full(8, 4.0), Positive()),
                            regularizer=Parameter(2.0, Positive()), dtype=dtype)
    basis = make()
    assert np.array_equal(basis.W[:, :8], g["W_head"])
    # (1) one evaluation at the reference's optimum
    one = SLM(make())
    one.obj_ = -np.inf
    one._state = one._make_state(X, y)
    try:
        nelbo, (ndvar, ndreg, ndhyp) = one._elbo(X, y, float(g["var_"]), float(g["reg_"]), g["hyp_"])
    finally:
        one._state.release()
        one._state = None
    assert abs(-nelbo - float(g["obj"])) < (1e-9 if dtype == "f64" else 1e-5) * abs(float(g["obj"]))
    assert normwise(one.weights_, g["m"]) < tol
    glog = np.concatenate(([ndvar * float(g["var_"])], [ndreg * float(g["reg_"])], np.asarray(ndhyp) * g["hyp_"]))
    assert np.abs(glog).max() < 1e-4 * abs(float(g["obj"])), glog
    # (2) the fit from the reference's start values
    slm = SLM(basis, var=Parameter(0.1, Positive()), nstarts=0, maxiter=500, random_state=0).fit(X, y)
    Ey, Vy = slm.predict_moments(Xs)
    ptol = max(1e-4, tol)
    assert abs(slm.obj_ - float(g["obj"])) < tol * abs(float(g["obj"]))
    assert abs(slm.var_ - float(g["var_"])) < ptol * float(g["var_"])
    assert abs(slm.regularizer_ - float(g["reg_"])) < ptol * float(g["reg_"])
    # the length scales of the two inputs that hardly matter (l = 8.08 and 9.07 against input weights 0.2 and 0.1) are the
    # flattest directions: the reference's own optimum moves by 0.8e-4 and 1.0e-4 there between two start points
    rel = np.abs(np.asarray(slm.hypers_) - g["hyp_"]) / g["hyp_"]
    flat = g["hyp_"] > 6.0
    assert flat.sum() == 2 and np.all(rel[~flat] < ptol) and np.all(rel[flat] < 3 * ptol), rel
    assert normwise(slm.weights_, g["m"]) < 20 * ptol
    ref_smse = float(g["smse"])
    assert abs(smse(g["ys_true"], g["Ey"]) - ref_smse) < 1e-12
    assert smse(g["ys_true"], Ey) <= 1.05 * ref_smse and np.all(Vy > 0)
    assert normwise(Ey, g["Ey"]) < 20 * ptol and normwise(Vy, g["Vy"]) < 20 * ptol


def test_fit_second_seed_ard_matern_vs_reference(golden):
    bs, Parameter, Positive, SLM = _imports()
    g = golden("fit_c1")
    X, y, Xs = g["s2_X"], g["s2_y"], g["s2_Xs"]
    basis = bs.RandomMatern32(nbases=20, Xdim=4, random_state=43, lenscale=Parameter(np.full(4, 1.3), Positive()),
                              regularizer=Parameter(2.0, Positive()))
    assert np.array_equal(basis.W, g["s2_W"])
    slm = SLM(basis, var=Parameter(0.4, Positive()), nstarts=0, maxiter=25, random_state=0).fit(X, y)
    Ey, Vy = slm.predict_moments(Xs)
    assert smse(g["s2_Ey"], Ey) < 5e-3
    assert abs(slm.obj_ - float(g["s2_obj"])) < 0.02 * abs(float(g["s2_obj"]))
    assert np.all(Vy > 0) and normwise(Vy, g["s2_Vy"]) < 0.3


@pytest.mark.timeout(900)
def test_hyper_gradient_f32_vs_f64_at_config2_size():
    """the f32 second pass forms A = Err (Phi_c m_s - Phi_s m_c) - (Phi_c U_s - Phi_s U_c) with
    cancellation; at N = 1M, F = 4096, D = 32 (config 2) its ARD hyper-gradient, objective and the other gradients
    are compared with the float64 path on the same GPU (f64 features, f64 MFMA Gram, f64 posterior, f64 second pass)."""
    bs, Parameter, Positive, SLM = _imports()
    N, d, n = 1_000_000, 32, 2048
    rng = np.random.default_rng(5)
    X = rng.standard_normal((N, d), dtype=np.float32)
    w = rng.standard_normal(d).astype(np.float32)
    y = (np.sin(X @ w / np.sqrt(d)) + 0.1 * rng.standard_normal(N, dtype=np.float32)).astype(np.float32)
    ls = np.linspace(0.8, 1.6, d)
    res = {}
    for dtype in ("f32", "f64"):
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()), dtype=dtype)
        slm = SLM(basis)
        slm.obj_ = -np.inf
        slm._state = basis.device_fit_state(X, y)
        try:
            f, (gv, gr, gh) = slm._elbo(X, y, 0.05, 2.0, ls)
        finally:
            slm._state.release()
            slm._state = None
        res[dtype] = (float(f), float(gv), float(gr), np.asarray(gh, dtype=float), slm.weights_.copy())
    f32, f64 = res["f32"], res["f64"]
    assert abs(f32[0] - f64[0]) < 1e-5 * abs(f64[0])
    assert abs(f32[1] - f64[1]) < 1e-3 * abs(f64[1]) and abs(f32[2] - f64[2]) < 1e-3 * abs(f64[2])
    assert normwise(f32[4], f64[4]) < 1e-3                  # posterior weights (north star: 1e-3 rel fp32)
    assert normwise(f32[3], f64[3]) < 2e-3, normwise(f32[3], f64[3])   # the ARD length-scale gradient


def test_dense_gram_float64_keeps_float64_arithmetic():
    """rr_dense_gram with float64 input used to pack to f32.  Unscaled linear features with a large offset
    lose their significant digits in f32; the float64 route (f64 MFMA SYRK) must not."""
    from revrand_amd import _hip
    rs = np.random.RandomState(0)
    for N, F in ((1000, 9), (4097, 130), (300, 300)):
        Phi = 1e4 + rs.randn(N, F)
        y = rs.randn(N)
        G, b, yty = _hip.dense_gram(Phi, y)
        Gr = Phi.T @ Phi
        assert normwise(G, Gr) < 1e-13 and np.array_equal(G, G.T)
        # the informative part: the covariance around the offset survives (f32 would leave ~1e-1 relative here)
        mu = Phi.mean(axis=0)
        assert normwise(G / N - np.outer(mu, mu), Gr / N - np.outer(mu, mu)) < 1e-6
        assert normwise(b, Phi.T @ y) < 1e-12 and abs(yty - y @ y) < 1e-12 * (y @ y)
    G32, _, _ = _hip.dense_gram(Phi.astype(np.float32))   # float32 in: f32 arithmetic, as before
    assert normwise(G32, Gr) < 1e-5


def test_dense_predict_vs_numpy():
    from revrand_amd import _hip
    rs = np.random.RandomState(1)
    for N, F in ((1, 1), (777, 9), (3000, 200)):
        Phi, m = rs.randn(N, F), rs.randn(F)
        A = rs.randn(F, F)
        C = A @ A.T / F
        Ey, Vf = _hip.dense_predict(Phi, m, C)
        assert normwise(Ey, Phi @ m) < 1e-12 and normwise(Vf, ((Phi @ C) * Phi).sum(axis=1)) < 1e-12


def test_concat_with_float64_child_uses_float64_statistics():
    """a float64 child (RandomLaplace with dtype="f64") + LinearBasis.  The f32 device feature matrix must
    not be used for the Gram while Err / m / gradients come from the f64 transform: `_elbo` against the f64 oracle."""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(2)
    N, d, n = 600, 3, 40
    X = rs.randn(N, d)
    y = np.sin(X @ np.array([1.0, -0.5, 0.3])) + 0.1 * rs.randn(N)
    cat = bs.RandomLaplace(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()), dtype="f64") \
        + bs.LinearBasis(onescol=True)
    assert cat.bases[0].dtype == "f64" and cat.gram(X, y, np.ones(d)) is None
    slm = SLM(cat)
    slm.obj_ = -np.inf
    ls = np.array([0.9, 1.2, 1.5])
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, 0.3, [1.5, 0.7], ls)
    W = cat.bases[0].W
    Phi = np.hstack((orc.rff_transform(X, W, ls), orc.linear_transform(X, True)))
    F = Phi.shape[1]
    dP = np.zeros((N, F, d))
    dP[:, :2 * n, :] = orc.rff_grad(X, W, ls)
    rd = np.concatenate((np.full(2 * n, 1.5), np.full(d + 1, 0.7)))
    o = orc.slm_elbo(Phi, y, 0.3, rd, [slice(0, 2 * n), slice(2 * n, F)], [dP[:, :, i] for i in range(d)])
    assert abs(-nelbo - o["elbo"]) < 1e-8 * abs(o["elbo"])
    assert normwise(slm.weights_, o["m"]) < 1e-7 and normwise(slm.covariance_, o["C"]) < 1e-7
    assert abs(-ndvar - o["dvar"]) < 1e-6 * abs(o["dvar"])
    assert normwise(-np.asarray(ndreg), o["dreg"]) < 1e-6
    assert normwise(-np.atleast_1d(ndhyp), np.array(o["dhyp"])) < 1e-5
    # predict_moments of such a concatenation: float64 on the GPU (rr_dense_predict), not a host matrix product
    slm.var_, slm.regularizer_, slm.hypers_ = 0.3, [1.5, 0.7], ls
    Xs = rs.randn(50, d)
    Ey, Vy = slm.predict_moments(Xs)
    Ps = np.hstack((orc.rff_transform(Xs, W, ls), orc.linear_transform(Xs, True)))
    Eo, Vo = orc.slm_predict_moments(Ps, o["m"], o["C"], 0.3)
    assert normwise(Ey, Eo) < 1e-7 and normwise(Vy, Vo) < 1e-7


def test_concat_with_float64_child_fits_resident_in_float64(monkeypatch):
    """a concatenation with a dtype="f64" child keeps (X, y) resident in a FLOAT64 feature matrix
    (rr_featmat64_*): f64 MFMA Gram, posterior, float64 second pass -- `_elbo` against the float64 oracle at the float64
    tolerances (north star: 1e-5 relative in fp64), host and device posterior, one chunk and several, an f32 child alongside,
    and bitwise reproducible in deterministic mode."""
    bs, Parameter, Positive, SLM = _imports()
    from revrand_amd import _hip
    from revrand_amd.basis_functions import CatFitState
    rs = np.random.RandomState(7)
    N, d, n = 1500, 5, 150
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    cat = bs.RandomLaplace(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()), dtype="f64") \
        + bs.LinearBasis(onescol=True) + bs.BiasBasis(offset=0.5) \
        + bs.RandomRBF(nbases=20, Xdim=d, random_state=4, lenscale=Parameter(1.0, Positive()))   # an f32 child, isotropic
    st = cat.device_fit_state(X, y)
    assert isinstance(st, CatFitState) and st.dtype == "f64" and type(st.fm).__name__ == "FeatureMatrix64"
    assert st.children[0].dX.dtype == np.float64 and st.children[1].dX.dtype == np.float64
    st.release()
    ls0, ls3 = np.linspace(0.8, 1.4, d), 1.3
    W0, W3 = cat.bases[0].W, cat.bases[3].W
    Phi = np.hstack((orc.rff_transform(X, W0, ls0), orc.linear_transform(X, True), np.full((N, 1), 0.5),
                     orc.rff_transform(X, W3, ls3)))
    F = Phi.shape[1]
    e = [0, 2 * n, 2 * n + d + 1, 2 * n + d + 2, F]
    dP0 = np.zeros((N, F, d))
    dP0[:, :2 * n, :] = orc.rff_grad(X, W0, ls0)
    dP3 = np.zeros((N, F))
    dP3[:, e[3]:] = orc.rff_grad(X, W3, ls3)       # isotropic: the reference's dimension-0-only gradient
    regs = [1.5, 0.7, 2.0, 1.1]
    rd = np.concatenate([np.full(e[i + 1] - e[i], regs[i]) for i in range(4)])
    o = orc.slm_elbo(Phi, y, 0.3, rd, [slice(e[i], e[i + 1]) for i in range(4)], [dP0[:, :, i] for i in range(d)] + [dP3])
    want_h = np.concatenate((o["dhyp"][:d], [o["dhyp"][d]]))
    dev = _hip.get_device()
    results = []
    for posdef, chunk, det in (("host", None, False), ("device", None, False), ("device", 384, False), ("device", 384, True),
                               ("device", 384, True)):
        monkeypatch.setenv("RR_POSDEF", posdef)
        prev = dev.set_deterministic(det)
        try:
            slm = SLM(cat)
            slm.obj_ = -np.inf
            kids = [b._resident_child(X, dtype="f64") for b in cat.bases]
            slm._state = CatFitState(cat, kids, X, y, chunk_rows=chunk, dtype="f64")
            nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, 0.3, regs, [ls0, ls3])
            C = slm._state.best_covariance() if slm._state.best_on_device else slm.covariance_
            slm._state.release()
        finally:
            dev.set_deterministic(prev)
        # float64 end to end; the posterior of this (Cauchy-frequency) model amplifies rounding by its condition number:
        # weights to 1e-6 where the north star asks for 1e-5
        assert abs(-nelbo - o["elbo"]) < 1e-9 * abs(o["elbo"]), (posdef, chunk)
        assert normwise(slm.weights_, o["m"]) < 1e-6 and normwise(C, o["C"]) < 1e-6, (posdef, chunk)
        assert abs(-ndvar - o["dvar"]) < 1e-7 * abs(o["dvar"])
        assert normwise(-np.asarray(ndreg), np.array(o["dreg"])) < 1e-7
        got_h = np.concatenate((np.atleast_1d(ndhyp[0]), [ndhyp[1]]))
        assert normwise(-got_h, want_h) < 1e-6, (posdef, chunk)
        results.append(np.concatenate(([nelbo, ndvar], np.asarray(ndreg), got_h, slm.weights_)))
    assert np.array_equal(results[3], results[4])          # deterministic mode: the same bits twice
    # a short fit through the estimator takes the float64 state by itself
    slm = SLM(cat, var=Parameter(0.3, Positive()), nstarts=0, maxiter=5, random_state=0).fit(X, y)
    assert np.isfinite(slm.obj_) and slm.weights_.shape == (F,)


def test_laplace_float64_phase_basis_takes_the_resident_f32_routes():
    """RandomLaplace (Cauchy W, phases of ~1e5 revolutions) in its default dtype -- the f32 pipeline with
    float64 phases (RR_F32P64) -- takes the resident concatenated fit state, and its `_elbo`, fused Gram and
    predict_moments match the float64 oracle at the f32 path's tolerance (1e-3; the plain f32 phase is 6e-2 off)."""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(2)
    N, d, n = 3000, 32, 192
    X = rs.randn(N, d)  # NOT float32-representable: the resident copy must stay float64
    y = np.sin(X @ rs.randn(d) / 3.0) + 0.1 * rs.randn(N)
    lap = bs.RandomLaplace(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()))
    assert lap.dtype == "f32" and lap.phase64 and np.abs(lap.W).max() > 1e3
    ls = np.linspace(0.8, 1.4, d)
    W = lap.W
    Phi_l = orc.rff_transform(X, W, ls)
    # fused single-basis Gram and the resident single-basis _elbo
    G, b, yty = lap.gram(X, y, ls)
    Go, bo, _ = orc.gram_stats(Phi_l, y)
    assert normwise(G, Go) < 1e-5 and normwise(b, bo) < 1e-5
    cat = lap + bs.LinearBasis(onescol=True)
    st = cat.device_fit_state(X, y)
    assert type(st).__name__ == "CatFitState" and st.children[0].dX.dtype == np.float64
    st.release()
    Gc, bc, _ = cat.gram(X, y, ls)
    Phi = np.hstack((Phi_l, orc.linear_transform(X, True)))
    assert normwise(Gc, Phi.T @ Phi) < 1e-5 and normwise(bc, Phi.T @ y) < 1e-5
    F = Phi.shape[1]
    dP = np.zeros((N, F, d))
    dP[:, :2 * n, :] = orc.rff_grad(X, W, ls)
    rd = np.concatenate((np.full(2 * n, 1.5), np.full(d + 1, 0.7)))
    o = orc.slm_elbo(Phi, y, 0.3, rd, [slice(0, 2 * n), slice(2 * n, F)], [dP[:, :, i] for i in range(d)])
    for basis, reg, ref in ((cat, [1.5, 0.7], o), (lap, 1.5, None)):
        slm = SLM(basis)
        slm.obj_ = -np.inf
        slm._state = slm._make_state(X, y)
        assert slm._state is not None
        nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, 0.3, reg, ls)
        slm._state.release()
        slm._state = None
        if ref is None:
            dPl = orc.rff_grad(X, W, ls)
            ref = orc.slm_elbo(Phi_l, y, 0.3, np.full(2 * n, 1.5), slice(None), [dPl[:, :, i] for i in range(d)])
        assert abs(-nelbo - ref["elbo"]) < 1e-5 * abs(ref["elbo"])
        assert normwise(slm.weights_, ref["m"]) < 1e-3
        assert abs(-ndvar - ref["dvar"]) < 1e-3 * abs(ref["dvar"])
        assert normwise(-np.atleast_1d(ndreg), np.array(ref["dreg"])) < 1e-3
        assert normwise(-np.atleast_1d(ndhyp), np.array(ref["dhyp"])) < 2e-3
    # predictions of the concatenation, covariance resident, variance as a sum of squares
    slm = SLM(cat)
    slm.var_, slm.regularizer_, slm.hypers_, slm.weights_, slm.covariance_ = 0.3, [1.5, 0.7], ls, o["m"], o["C"]
    Xs = rs.randn(300, d)
    Ey, Vy = slm.predict_moments(Xs)
    Ps = np.hstack((orc.rff_transform(Xs, W, ls), orc.linear_transform(Xs, True)))
    Eo, Vo = orc.slm_predict_moments(Ps, o["m"], o["C"], 0.3)
    assert normwise(Ey, Eo) < 1e-3 and normwise(Vy, Vo) < 1e-3
    assert normwise(slm.predict(Xs), Eo) < 1e-3
    # the contraction entry point (glm.py:274-275) on the same basis
    E = rs.randn(200, 2 * n)
    dPl = orc.rff_grad(X[:200], W, ls)
    want = np.array([(E * dPl[:, :, i]).sum() for i in range(d)])
    assert normwise(lap.grad_contract(X[:200], E, ls), want) < 1e-3


def test_default_linear_basis_model_predicts_on_device():
    """StandardLinearModel() (LinearBasis, slm.py:57) has no fused route: Gram and predict_moments still run on the GPU,
    in float64, and equal the host formulas on data with a large offset."""
    bs, Parameter, Positive, SLM = _imports()
    rs = np.random.RandomState(4)
    X = 500.0 + rs.randn(400, 2)
    y = 3.0 + X @ np.array([0.5, -0.25]) + 0.01 * rs.randn(400)
    slm = SLM(bs.LinearBasis(onescol=True), nstarts=0, maxiter=50, random_state=0).fit(X, y)
    Xs = 500.0 + rs.randn(30, 2)
    Ey, Vy = slm.predict_moments(Xs)
    Ps = orc.linear_transform(Xs, True)
    Eo, Vo = orc.slm_predict_moments(Ps, slm.weights_, slm.covariance_, slm.var_)
    assert normwise(Ey, Eo) < 1e-10 and normwise(Vy, Vo) < 1e-8
    assert smse(3.0 + Xs @ np.array([0.5, -0.25]), Ey) < 1e-2


@pytest.mark.timeout(900)
def test_config3_full_one_gpu_share_properties():
    """BASELINE configs[2] at one GPU's full share (N = 10M / 8 = 1.25M rows, RandomMatern52 n=4096 + LinearBasis, D=64,
    F_tot = 8257): size-independent properties of the device-side concatenation + Gram -- the oracle cannot run this size."""
    bs, Parameter, Positive, _ = _imports()
    N, d, n = 1_250_000, 64, 4096
    rng = np.random.default_rng(11)
    X = rng.standard_normal((N, d), dtype=np.float32)
    y = rng.standard_normal(N, dtype=np.float32)
    cat = bs.RandomMatern52(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True)
    hyp = [np.linspace(0.7, 1.9, d)]
    st = cat.device_fit_state(X, y)
    try:
        yty = st.gram_device(hyp)
        G, b, _ = st.stats_host()
    finally:
        st.release()
    F = 2 * n + d + 1
    assert G.shape == (F, F) and np.array_equal(G, G.T)
    # cos^2 + sin^2 = 1 per frequency: trace of the random Fourier block is N
    assert abs(np.trace(G[:2 * n, :2 * n]) - N) < 1e-5 * N
    # the LinearBasis block is [1, X]^T [1, X]: exact count, column sums and second moments against float64 NumPy
    X64 = X.astype(np.float64)
    assert G[2 * n, 2 * n] == N
    assert normwise(G[2 * n, 2 * n + 1:], X64.sum(axis=0)) < 1e-5
    assert normwise(G[2 * n + 1:, 2 * n + 1:], X64.T @ X64) < 1e-5
    assert normwise(b[2 * n:], np.concatenate(([y.astype(np.float64).sum()], X64.T @ y.astype(np.float64)))) < 1e-5
    assert abs(yty - float((y.astype(np.float64) ** 2).sum())) < 1e-6 * N
    # a slice of the cross block against the oracle's features of the first rows would need all rows; instead: additivity
    # over row shards (what the multi-GPU exchange relies on) at full size
    h = N // 2
    parts = []
    for lo, hi in ((0, h), (h, N)):
        s2 = cat.device_fit_state(X[lo:hi], y[lo:hi])
        try:
            s2.gram_device(hyp)
            parts.append(s2.stats_host())
        finally:
            s2.release()
    assert normwise(parts[0][0] + parts[1][0], G) < 2e-6 and normwise(parts[0][1] + parts[1][1], b) < 1e-5


@pytest.mark.timeout(900)
def test_config4_full_size_stream_properties():
    """BASELINE configs[3] at full size: FastFoodRBF nbases=8192, D=128 (F=16384), N = 4M rows streamed through a two-slot
    device ring in 16 chunks of 262 144 rows (Phi would be 262 GB).  Every chunk: unit row norm (sum_j Phi_j^2 = 1) on
    sampled rows and exact agreement with the oracle chain on a few of them."""
    from revrand_amd import _hip
    bs, _, _, _ = _imports()
    d, nb, CH, NCH = 128, 8192, 262_144, 16
    f = bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=1)
    h = f._handles()[0]
    F = 2 * h.n
    assert F == 16384
    dev = _hip.get_device()
    rng = np.random.default_rng(12)
    base = rng.standard_normal((CH, d), dtype=np.float32)
    ring = [dev.malloc(CH * F * 4) for _ in range(2)]
    dX = [dev.upload_matrix(base), dev.upload_matrix(base)]
    rows = rng.integers(0, CH, size=48)
    try:
        for c in range(NCH):
            scale = np.float32(1.0 + 0.05 * c)          # chunk c of the stream: a rescaled copy (generation is not the test)
            Xc = base * scale
            dev.upload_rows(dX[c & 1], 0, Xc)
            h.transform_dev(dX[c & 1], 1.3, ring[c & 1], np.float32)
            dev.sync()
            out = np.stack([dev.download(ring[c & 1], (F,), np.float32, offset_bytes=int(r) * F * 4) for r in rows])
            assert np.abs((out.astype(np.float64) ** 2).sum(axis=1) - 1.0).max() < 1e-4
            if c in (0, NCH - 1):
                ref = orc.fastfood_transform(Xc[rows[:8]].astype(np.float64), f.B, f.G, f.PI, f.S, 1.3)
                assert normwise(out[:8], ref) < 1e-3
    finally:
        for bfr in ring + dX:
            bfr.free()


@pytest.mark.parametrize("kind", ["rff", "concat"])
def test_predictive_variance_is_a_sum_of_squares_for_badly_scaled_covariances(kind):
    """predict_moments' variance phi^T C phi (slm.py:242-243) in float32 loses its digits when C is badly scaled: the error
    goes with |phi|^T |C| |phi|.  A fitted estimator therefore factors its covariance once (C = M M^T, float64, on the
    device) and forms || phi^T M ||^2.  Here: the posterior of a vague prior and little noise; the estimator's variance must
    match float64 NumPy to 1e-3 (the float32 quadratic form on the same inputs is printed next to it)."""
    bs, Parameter, Positive, SLM = _imports()
    from revrand_amd import _hip
    rs = np.random.RandomState(0)
    d, n = 4, 60
    if kind == "rff":
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    else:
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) \
            + bs.LinearBasis(onescol=True)
    Xs = rs.randn(700, d)
    ls = np.linspace(0.8, 1.4, d)
    Phi = basis.transform(Xs, ls)
    F = Phi.shape[1]
    # a posterior as a fit with a vague prior and little noise leaves it: C = (I / L + G / var)^-1 with L = 1e6, var = 1e-3.
    # Directions the training features barely excite keep variances near L, the others fall to var / N: query features
    # from the same distribution sit almost entirely in the latter, and phi^T C phi is the small difference of big terms.
    Ptr = basis.transform(rs.randn(4000, d), ls)
    G = Ptr.T @ Ptr
    C = np.linalg.inv(np.eye(F) / 1e6 + G / 1e-3)
    C = 0.5 * (C + C.T)
    m = rs.randn(F)
    slm = SLM(basis)
    slm.var_, slm.regularizer_, slm.hypers_, slm.weights_, slm.covariance_ = 0.0, 1.0, ls, m, C
    Ey, Vy = slm.predict_moments(Xs)
    Eo, Vo = Phi @ m, ((Phi @ C) * Phi).sum(axis=1)
    assert normwise(Ey, Eo) < 1e-4
    assert np.all(np.abs(Vy - Vo) <= 1e-3 * Vo), float(np.abs(Vy / Vo - 1).max())
    cov = slm._device_covariance()
    assert isinstance(cov, _hip.DeviceCovariance) and cov.factor()[1] == 1
    # the float32 quadratic form on the same inputs, for the record (host C -> triangular float32 form)
    _, Vq = basis.predict_moments(Xs, ls, m, C)
    print("max relative error of the variance: sum of squares %.2e, float32 quadratic form %.2e"
          % (np.abs(Vy / Vo - 1).max(), np.abs(Vq / Vo - 1).max()))
    assert np.abs(Vq / Vo - 1).max() > np.abs(Vy / Vo - 1).max()
    # a covariance that is not positive definite (as the SVD route can return) keeps the quadratic form, and still works
    Cn = C.copy()
    Cn[0, 0] = -1.0
    slm.covariance_ = Cn
    _, Vn = slm.predict_moments(Xs)
    assert slm._device_covariance().factor()[1] == 0
    assert normwise(Vn, ((Phi @ Cn) * Phi).sum(axis=1)) < 5e-2


@pytest.mark.parametrize("n", [257, 288, 289, 321, 352, 353, 400])
def test_gram_with_a_ragged_last_column_block(n):
    """F = 2n not a multiple of 256: the tiles of the last column block run in rr_syrk_f32_ragged_kernel (transposed,
    only the 32-column blocks that hold valid columns) when it has <= 192 valid columns.  Widths on both sides of every
    boundary (2, 64, 66, 130, 192, 194 valid columns; 800 = 3 blocks + 32), ragged row counts, vs the float64 oracle."""
    bs, Parameter, Positive, _ = _imports()
    rs = np.random.RandomState(n)
    N, d = 3000 + n, 7
    X = rs.randn(N, d).astype(np.float32)
    y = rs.randn(N).astype(np.float32)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=2)
    G, bv, yty = b.gram(X, y, 1.1)
    Gr, br, tr = orc.rff_gram_chunked(X.astype(np.float64), y.astype(np.float64), b.W, 1.1)
    assert np.array_equal(G, G.T)
    assert normwise(G, Gr) < 2e-5 and normwise(bv, br) < 1e-4 and abs(yty - tr) < 1e-6 * tr
    # the last block's columns specifically (a transposed flush that mixes up rows and columns shows here)
    c0 = (2 * n - 1) // 256 * 256
    assert normwise(G[:, c0:], Gr[:, c0:]) < 2e-5
