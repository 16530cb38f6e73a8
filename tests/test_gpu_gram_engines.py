"""
The split engines of the f32 Gram (RR_GRAM_BF16X3 / RR_GRAM_BF16X4 / RR_GRAM_FP16X3, include/revrand_hip.h): f32 feature
values as 16-bit hi + lo, 3 or 4 16-bit products per f32 product on the bf16/fp16 matrix pipe, f32 accumulation.  Parity against the
NumPy oracle with the tolerance of the f32 path (1e-3 relative, BASELINE.json) and against the measured accuracy of the
engines themselves (a few 1e-6 of max|G|), for every producer of the feature matrix: the MFMA feature kernel writing
the K-blocked layout directly, and the conversion kernel behind concatenated bases and Xdim > 128.
"""
import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise

pytestmark = pytest.mark.gpu


def _imports():
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd import _hip
    return bs, Parameter, Positive, _hip


@pytest.fixture(params=["bf16x3", "bf16x4", "fp16x3"])
def engine(request):
    from revrand_amd import _hip
    dev = _hip.get_device()
    prev = dev.set_gram_engine(request.param)
    assert dev.gram_engine == request.param
    yield request.param
    dev.set_gram_engine(prev)


ENGINE_TOL = {"bf16x3": 1e-5, "bf16x4": 6e-6, "fp16x3": 2.5e-6}   # measured: <= 5e-6 / 2.5e-6 / 1.1e-6 of max|G|
# fp16x3 (random Fourier features scaled into [-1, 1], fp16 hi + lo: 22 mantissa bits) matches the f32 MFMA engine's own
# error against the float64 oracle; it applies to feature matrices written by the MFMA feature kernel, everything else
# (concatenations, Xdim > 128, U = Phi C) runs bf16x3 under that setting.


@pytest.mark.parametrize("shape", [(1, 3, 4), (64, 4, 16), (500, 4, 16), (4099, 8, 128), (3000, 32, 200), (1500, 21, 260),
                                   (70000, 8, 128), (1000, 64, 128), (777, 100, 40)])
def test_gram_vs_oracle(engine, shape):
    bs, Parameter, Positive, _ = _imports()
    N, d, n = shape
    rs = np.random.RandomState(N + n)
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)).astype(np.float32)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive()))
    ls = np.linspace(0.8, 1.6, d)
    G, bv, yty = b.gram(X, y, ls)
    Gr, br, ytyr = orc.rff_gram_chunked(X.astype(np.float64), y.astype(np.float64), b.W, ls)
    assert G.shape == (2 * n, 2 * n) and np.array_equal(G, G.T)
    assert normwise(G, Gr) < ENGINE_TOL[engine], normwise(G, Gr)
    assert normwise(bv, br) < 1e-5            # Phi^T y is accumulated from the f32 values, before the split
    assert abs(yty - ytyr) < 1e-5 * ytyr
    G2, b2, _ = b.gram(X.astype(np.float64), None, ls)      # float64 X, no y
    assert b2 is None and normwise(G2, Gr) < ENGINE_TOL[engine]


def test_row_chunks_and_engines_agree(engine, monkeypatch):
    """Several row chunks (pad rows at each seam are zero in the K-blocked layout) and the three engines on one input."""
    bs, Parameter, Positive, _hip = _imports()
    N, d, n = 5003, 16, 300
    rs = np.random.RandomState(0)
    X = rs.randn(N, d).astype(np.float32)
    y = rs.randn(N).astype(np.float32)
    b = bs.RandomMatern32(nbases=n, Xdim=d, random_state=1)
    G1, b1, _ = b.gram(X, y, 1.2)
    monkeypatch.setenv("RR_GRAM_CHUNK_ROWS", "1024")
    Gc, bc, _ = b.gram(X, y, 1.2)
    monkeypatch.delenv("RR_GRAM_CHUNK_ROWS")
    assert normwise(Gc, G1) < 2e-6 and normwise(bc, b1) < 1e-6      # f32 partial sums over different row groups
    dev = _hip.get_device()
    dev.set_gram_engine("f32")
    Gf, bf, _ = b.gram(X, y, 1.2)
    dev.set_gram_engine(engine)
    assert normwise(G1, Gf) < ENGINE_TOL[engine] and normwise(b1, bf) < 1e-6


def test_linearity_over_row_shards(engine):
    bs, Parameter, Positive, _ = _imports()
    N, d, n = 40000, 32, 512
    rs = np.random.RandomState(3)
    X = rs.randn(N, d).astype(np.float32)
    y = rs.randn(N).astype(np.float32)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=7)
    G, bv, yty = b.gram(X, y, 1.0)
    Ga, ba, ta = b.gram(X[:17001], y[:17001], 1.0)
    Gb, bb, tb = b.gram(X[17001:], y[17001:], 1.0)
    assert normwise(Ga + Gb, G) < 3e-6 and normwise(ba + bb, bv) < 1e-6 and abs(ta + tb - yty) < 1e-9 * yty
    dg = np.diag(G)
    assert np.abs(dg[:n] + dg[n:] - N / n).max() < 2e-5 * (N / n)      # cos^2 + sin^2 = 1 per frequency
    assert abs(np.trace(G) - N) < 1e-5 * N


def test_concatenated_bases_and_wide_inputs_use_the_conversion_kernel(engine):
    """Feature matrices not written by the MFMA feature kernel (device-side concatenation; Xdim > 128) are converted
    from row-major f32 to the K-blocked bf16 layout before the same SYRK."""
    bs, Parameter, Positive, _ = _imports()
    rs = np.random.RandomState(8)
    N, d, n = 3001, 12, 150
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    ls = np.linspace(0.7, 1.4, d)
    base = bs.RandomMatern52(nbases=n, Xdim=d, random_state=4, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True) + bs.BiasBasis(offset=2.0)
    G, b, yty = base.gram(X, y, ls)
    ref = np.hstack((orc.rff_transform(X, base.bases[0].W, ls), orc.linear_transform(X), np.full((N, 1), 2.0)))
    assert np.array_equal(G, G.T)
    assert normwise(G, ref.T @ ref) < 2e-5 and normwise(b, ref.T @ y) < 1e-4
    Xw = rs.randn(1500, 200) / 5.0
    yw = rs.randn(1500)
    bw = bs.RandomRBF(nbases=130, Xdim=200, random_state=2)
    Gw, bvw, _ = bw.gram(Xw.astype(np.float32), yw.astype(np.float32), 1.3)
    Gr, br, _ = orc.rff_gram_chunked(Xw.astype(np.float32).astype(np.float64), yw.astype(np.float32).astype(np.float64), bw.W, 1.3)
    assert normwise(Gw, Gr) < 2e-5 and normwise(bvw, br) < 1e-4


@pytest.mark.parametrize("tag", ["iso", "ard"])
def test_resident_elbo_matches_reference(golden, engine, tag):
    """The whole `_elbo` (Gram on the split-bf16 engine, device posterior / host Cholesky, second pass) against the
    reference's golden ELBO, gradients and posterior weights, at the tolerances of the f32 path."""
    bs, Parameter, Positive, _ = _imports()
    from revrand_amd import StandardLinearModel as SLM
    g = golden("elbo")
    X, y = g["X"], g["y"]
    d, n = X.shape[1], 16
    lsp = Parameter(1., Positive()) if tag == "iso" else Parameter(np.ones(d), Positive())
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=21, lenscale=lsp)
    slm = SLM(basis)
    slm.obj_ = -np.inf
    slm._state = basis.device_fit_state(X, y)
    ls = float(g[tag + "_ls"]) if tag == "iso" else g[tag + "_ls"]
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, float(g["var"]), float(g["reg"]), ls)
    slm._state.release()
    assert abs(-nelbo - g[tag + "_elbo"]) < 1e-4 * abs(g[tag + "_elbo"])
    assert normwise(slm.weights_, g[tag + "_m"]) < 1e-3
    assert normwise(-ndvar, g[tag + "_dvar"]) < 1e-3
    assert normwise(-np.atleast_1d(ndreg), g[tag + "_dreg"]) < 1e-3
    assert normwise(-np.atleast_1d(ndhyp), g[tag + "_dhyp"]) < 2e-3


@pytest.mark.parametrize("shape", [(700, 5, 40), (1500, 16, 96), (1025, 21, 130), (2000, 6, 300)])
def test_second_pass_and_predict_vs_oracle(engine, shape):
    """U = Phi C of the second pass / predict_moments runs on the same split-bf16 engine (GEMM mode of the kernel)."""
    bs, Parameter, Positive, _ = _imports()
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
    assert normwise(dh, -np.array(o["dhyp"])) < 2e-3
    Xs = rs.randn(257, d)
    Ey, Vf = basis.predict_moments(Xs, ls, o["m"], o["C"])
    Eo, Vo = orc.slm_predict_moments(orc.rff_transform(Xs, basis.W, ls), o["m"], o["C"], 0.0)
    assert normwise(Ey, Eo) < 1e-4 and normwise(Vf, Vo) < 1e-3


def test_posterior_weights_within_tolerance(engine):
    """North star: 1e-3 relative on the posterior weights of the f32 path."""
    bs, Parameter, Positive, _ = _imports()
    N, d, n = 20000, 8, 128
    rs = np.random.RandomState(5)
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)).astype(np.float32)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=3)
    G, bv, _ = b.gram(X, y, 1.1)
    Gr, br, _ = orc.rff_gram_chunked(X.astype(np.float64), y.astype(np.float64), b.W, 1.1)
    var, reg = 0.05, 1.0
    m = np.linalg.solve(np.eye(2 * n) / reg + G / var, bv / var)
    mr = np.linalg.solve(np.eye(2 * n) / reg + Gr / var, br / var)
    assert normwise(m, mr) < 1e-3, normwise(m, mr)


def test_engine_api():
    _, _, _, _hip = _imports()
    dev = _hip.get_device()
    prev = dev.gram_engine
    with pytest.raises(ValueError):
        dev.set_gram_engine("fp8")
    assert dev.set_gram_engine("bf16x3") == prev and dev.gram_engine == "bf16x3"
    rc = dev.lib.rr_set_gram_engine(dev.ctx, 7)
    assert rc != 0 and dev.gram_engine == "bf16x3"
    dev.set_gram_engine(prev)
    assert dev.gram_engine == prev


def test_f64_gram_is_unaffected(engine):
    bs, Parameter, Positive, _ = _imports()
    rs = np.random.RandomState(1)
    X, y = rs.randn(900, 5), rs.randn(900)
    b = bs.RandomRBF(nbases=40, Xdim=5, random_state=2, dtype="f64")
    G, bv, _ = b.gram(X, y, 0.9)
    Gr, br, _ = orc.rff_gram_chunked(X, y, b.W, 0.9)
    assert normwise(G, Gr) < 1e-12 and normwise(bv, br) < 1e-12


@pytest.mark.parametrize("chunk_rows", [None, 512])
def test_concat_second_pass_and_predict_on_the_engine(engine, chunk_rows, monkeypatch):
    """The concatenated-basis case of test_gpu_slm.py (statistics, sqErr, per-child gradients, predict_moments) with the
    Gram and U = Phi C on the split-bf16 engine (whose second pass keeps the stored route at every width)."""
    from test_gpu_slm import test_concat_second_pass_and_predict_vs_oracle as concat_case
    concat_case(monkeypatch, chunk_rows, 70, 90, None)
    if chunk_rows is None:
        concat_case(monkeypatch, None, 256, 512, None)


def test_glm_step_on_the_engine(engine, golden):
    """With a split engine selected the three GEMMs of the SVI step (fs = ws Phi^T, dfs Phi with a K-split over the
    rows, dfs^T ws) run on the 16-bit matrix pipe: the reference's golden minibatch ELBO / gradients, the concatenated
    case and the config-5-sized additivity check at their usual tolerances."""
    import test_gpu_glm as tg
    for tag, lik in tg.CASES:
        tg.test_minibatch_elbo_vs_reference(golden, tag, lik)
    tg.test_minibatch_elbo_concat_and_generic_children_vs_oracle()
    tg.test_config5_full_minibatch_is_additive_over_rows()


def test_estimator_keyword_selects_the_engine_for_fit_and_predict():
    """`gram_engine=` on the estimators scopes the setting to fit / predict_moments and survives sklearn's clone."""
    from sklearn.base import clone
    bs, Parameter, Positive, _hip = _imports()
    from revrand_amd import StandardLinearModel, GeneralizedLinearModel
    import revrand_amd.likelihoods as lk
    dev = _hip.get_device()
    assert dev.gram_engine == "f32"
    rs = np.random.RandomState(0)
    X = rs.randn(4000, 3)
    y = np.sin(X[:, 0]) + 0.1 * rs.randn(4000)
    seen = []
    orig = bs.DeviceFitState.gram_device

    def spy(self, *a, **k):
        seen.append(dev.gram_engine)
        return orig(self, *a, **k)
    bs.DeviceFitState.gram_device = spy
    try:
        basis = bs.RandomRBF(nbases=150, Xdim=3, random_state=1, lenscale=Parameter(1.0, Positive()),
                             regularizer=Parameter(1.0, Positive()))
        slm = StandardLinearModel(basis, var=Parameter(0.5, Positive()), maxiter=30, nstarts=0, gram_engine="fp16x3")
        assert clone(slm).get_params()["gram_engine"] == "fp16x3"
        slm.fit(X, y)
    finally:
        bs.DeviceFitState.gram_device = orig
    assert seen and set(seen) == {"fp16x3"} and dev.gram_engine == "f32"
    Ey, Vy = slm.predict_moments(X[:500])
    assert ((y[:500] - Ey) ** 2).mean() / y[:500].var() < 0.1 and dev.gram_engine == "f32"
    glm = GeneralizedLinearModel(lk.Gaussian(), bs.RandomRBF(nbases=50, Xdim=3, random_state=1), maxiter=40, batch_size=500,
                                 nstarts=0, random_state=1, gram_engine="bf16x3")
    glm.fit(X, y)
    assert dev.gram_engine == "f32" and np.isfinite(glm.predict(X[:50])).all()
    with pytest.raises(ValueError):
        StandardLinearModel(bs.RandomRBF(nbases=10, Xdim=3), gram_engine="int4").fit(X, y)


def test_random_shapes_agree_with_the_f32_engine():
    """40 random (N, d, n): ragged rows (pad rows of the K-blocked layout), partial column blocks, d = 1..128, with and
    without y -- each split engine against the f32 MFMA engine on the same input."""
    bs, Parameter, Positive, _hip = _imports()
    dev = _hip.get_device()
    rs = np.random.RandomState(123)
    # worst case per product (nothing averages at N = 1): bf16x3 3 * 2^-18, bf16x4 2 * 2^-18, fp16x3 ~2^-21, plus the
    # f32 engine's own rounding
    tol = {"fp16x3": 3e-6, "bf16x3": 2.5e-5, "bf16x4": 1.5e-5}
    try:
        for it in range(40):
            N = int(rs.choice([1, 31, 63, 64, 65, 257, 1000, 4097, 9000]))
            d = int(rs.choice([1, 2, 7, 8, 9, 16, 31, 33, 64, 100, 128]))
            n = int(rs.choice([1, 5, 16, 31, 32, 33, 100, 128, 129, 300]))
            X = rs.randn(N, d).astype(np.float32 if it % 2 else np.float64)
            y = rs.randn(N) if it % 3 else None
            b = bs.RandomMatern52(nbases=n, Xdim=d, random_state=it)
            dev.set_gram_engine("f32")
            Gf, bf, _ = b.gram(X, y, 1.1)
            for eng in ("fp16x3", "bf16x3", "bf16x4"):
                dev.set_gram_engine(eng)
                G, bv, _ = b.gram(X, y, 1.1)
                assert np.array_equal(G, G.T)
                assert normwise(G, Gf) < tol[eng], (eng, N, d, n, normwise(G, Gf))
                assert (bv is None) == (y is None)
                if y is not None:
                    assert normwise(bv, bf) < 1e-5, (eng, N, d, n)
    finally:
        dev.set_gram_engine("f32")


def test_conversion_path_beyond_65535_k_steps():
    """1.1 M rows through the conversion kernel (Xdim > 128 -> f32 features -> K-blocked bf16): more 16-row k-steps than
    a grid's y dimension holds."""
    bs, Parameter, Positive, _hip = _imports()
    dev = _hip.get_device()
    N, d, n = 1_100_003, 130, 40
    rng = np.random.default_rng(0)
    X = (rng.standard_normal((N, d), dtype=np.float32) / 4).astype(np.float32)
    y = rng.standard_normal(N, dtype=np.float32)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=3)
    Gf, bf, _ = b.gram(X, y, 1.0)
    prev = dev.set_gram_engine("bf16x3")
    try:
        G, bv, _ = b.gram(X, y, 1.0)
    finally:
        dev.set_gram_engine(prev)
    assert normwise(G, Gf) < 1e-5 and normwise(bv, bf) < 1e-6
    assert abs(np.trace(G) - N) < 1e-5 * N
