"""The SVI loop of GeneralizedLinearModel.fit with its parameters resident in HBM (rr_glm_sgd, glm._ResidentLoop) against
the host loop it replaces -- optimize.sgd o logtrick_sgd o structured_sgd around `_elbo` (reference: optimize/sgd.py:337-425,
optimize/decorators.py:133-252,329-408, glm.py:205-294), itself held to the reference's golden `_elbo` outputs by
test_gpu_glm.py.  Same seeds -> same minibatches, same draws, same start point: the two loops must produce the same fit."""
import logging

import numpy as np
import pytest

from conftest import normwise

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["two streams", "one stream"])
def _order_of_work(monkeypatch, request):
    """Both forms of the loop are held to the host loop by every test here: with the second stream and feature matrix (next
    step's features under this step's Ed product -- the library's choice when a step is large, rows x F >= 2^22: config 5's
    shape below) and without (small steps)."""
    monkeypatch.setenv("RR_GLM_SGD_OVERLAP", "1" if request.param == "two streams" else "0")


def _imports():
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd import optimize as opt
    from revrand_amd.btypes import Bound, Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    return bs, lk, opt, Bound, Parameter, Positive, GeneralizedLinearModel


def _data(lik, N=6000, d=5, seed=4):
    rs = np.random.RandomState(seed)
    X = rs.randn(N, d)
    f = 0.5 * np.sin(X[:, 0]) + 0.2 * X[:, 2]
    if lik == "poisson" or lik == "poisson_softplus":
        return X, rs.poisson(np.exp(f)).astype(float), ()
    if lik == "bernoulli":
        return X, (rs.rand(N) < 1 / (1 + np.exp(-3 * f))).astype(float), ()
    if lik == "binomial":
        n = rs.randint(5, 30, size=N).astype(float)
        return X, rs.binomial(n.astype(int), 1 / (1 + np.exp(-3 * f))).astype(float), (n,)
    return X, f + 0.1 * rs.randn(N), ()


def _fit(resident, lik="poisson", updater=None, iso=False, sampler="host", maxiter=20, K=3, L=8, batch=1500, nstarts=2,
         lenscale=None, monkeypatch=None, nbases=64, count=None):
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    from revrand_amd import _hip
    X, y, largs = _data(lik)
    d = X.shape[1]
    if lenscale is None:
        lenscale = Parameter(1.0, Positive()) if iso else Parameter(np.ones(d), Positive())
    basis = bs.RandomRBF(nbases=nbases, Xdim=d, random_state=1, lenscale=lenscale)
    like = {"poisson": lambda: lk.Poisson(), "poisson_softplus": lambda: lk.Poisson("softplus"), "bernoulli": lk.Bernoulli,
            "binomial": lk.Binomial, "gaussian": lk.Gaussian}[lik]()
    glm = GLM(like, basis, K=K, nsamples=L, batch_size=batch, maxiter=maxiter, nstarts=nstarts, random_state=11,
              updater=updater() if updater is not None else None, sampler=sampler)
    glm._resident_sgd = resident
    steps = [0]
    if monkeypatch is not None:
        real = _hip.ResidentSgd.step

        def spy(self, *a, **k):
            steps[0] += 1
            return real(self, *a, **k)
        monkeypatch.setattr(_hip.ResidentSgd, "step", spy)
    np.random.seed(3)  # (the start point is a draw from NumPy's global stream, as in the reference)
    glm.fit(X, y, likelihood_args=largs)
    if count is not None:
        count.append(steps[0])
    return (glm.weights_.copy(), glm.covariance_.copy(), np.atleast_1d(np.array(glm.regularizer_, dtype=float)),
            np.atleast_1d(np.array(glm.like_hypers_, dtype=float)), np.atleast_1d(np.array(glm.basis_hypers_, dtype=float)),
            glm.random_.randn())


def _same(a, b, tol):
    for u, v in zip(a[:5], b[:5]):
        assert u.shape == v.shape
        if u.size:
            assert normwise(u, v) < tol, (normwise(u, v), tol)
    assert a[5] == b[5]  # the RandomState ends in the same state: same minibatches, same draws consumed


@pytest.mark.parametrize("lik", ["poisson", "poisson_softplus", "bernoulli", "binomial", "gaussian"])
def test_resident_loop_equals_host_loop(lik, monkeypatch):
    """20 Adam steps from the same start on the same minibatches with the same draws: every fitted block agrees (the step's
    float32 K-split atomics allow last-bit differences per step, as between two host-loop runs)."""
    count = []
    dev = _fit(True, lik, monkeypatch=monkeypatch, count=count)
    assert count == [20]  # every step went through rr_glm_sgd_step ...
    monkeypatch.undo()
    host = _fit(False, lik, monkeypatch=monkeypatch, count=count)
    assert count[1] == 0   # ... and none of the host loop's did
    _same(dev, host, 2e-5)


def test_isotropic_length_scale_takes_dimension_zero_only_like_the_reference():
    _same(_fit(True, iso=True), _fit(False, iso=True), 2e-5)


@pytest.mark.parametrize("name", ["SGDUpdater", "AdaDelta", "AdaGrad", "Momentum", "Adam"])
def test_every_updater_of_the_reference(name):
    """sgd.py:14-330: the five update rules, state in HBM."""
    opt = _imports()[2]
    # (AdaGrad's default eta = 1 makes the first step a unit step along sign(grad) in every coordinate: chaos, not arithmetic)
    mk = {"SGDUpdater": lambda: opt.SGDUpdater(eta=1e-4), "Momentum": lambda: opt.Momentum(rho=0.5, eta=1e-4),
          "AdaGrad": lambda: opt.AdaGrad(eta=1e-2)}.get(name, getattr(opt, name))
    _same(_fit(True, updater=mk, maxiter=12), _fit(False, updater=mk, maxiter=12), 2e-5)


def test_three_steps_agree_to_the_noise_of_the_steps_float32_atomics():
    """Three Adam steps: the loops' own arithmetic is float64 on both sides (they differ in the order of the sums of the
    mixture terms and the regulariser gradient, 1e-16); what is left is the last-bit noise of the step's float32 K-split
    atomics (1e-7 of a gradient entry) times Adam's step length 0.01: 1e-9 of the parameters -- measured 7e-10."""
    _same(_fit(True, maxiter=3), _fit(False, maxiter=3), 1e-8)


def test_bounded_coordinates_are_truncated_and_clipped():
    """A length scale with a plain Bound (no log trick) that the optimiser pushes into its bound: sgd.py:404-420."""
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    mk = lambda: Parameter(np.full(5, 1.0), Bound(0.97, 1.02))  # noqa: E731
    a, b = _fit(True, lenscale=mk(), maxiter=15), _fit(False, lenscale=mk(), maxiter=15)
    _same(a, b, 2e-5)
    assert np.all(a[4] >= 0.97) and np.all(a[4] <= 1.02) and (np.any(a[4] == 0.97) or np.any(a[4] == 1.02))


def test_device_sampler_and_the_log_lines(caplog):
    """sampler="device": same fit from both loops (the draws are a function of (seed, step)); the `Iter n: ELBO` lines of
    glm.py:287-290 come from the device's objective."""
    with caplog.at_level(logging.INFO, logger="revrand_amd.glm"):
        a = _fit(True, sampler="device", maxiter=10)
    dev_lines = [r.getMessage() for r in caplog.records if r.getMessage().startswith("Iter ")]
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="revrand_amd.glm"):
        b = _fit(False, sampler="device", maxiter=10)
    host_lines = [r.getMessage() for r in caplog.records if r.getMessage().startswith("Iter ")]
    _same(a, b, 2e-5)
    assert len(dev_lines) == len(host_lines) == 2  # iterations 0 and maxiter - 1

    def elbo(line):
        return float(line.split("ELBO = ")[1].split(",")[0])
    for u, v in zip(dev_lines, host_lines):
        assert u.split(":")[0] == v.split(":")[0]
        assert abs(elbo(u) - elbo(v)) < 1e-5 * abs(elbo(v))


@pytest.mark.parametrize("lik", ["poisson", "gaussian", "binomial"])
def test_concatenation_of_fourier_and_linear_children(lik, monkeypatch):
    """The reference's own model tests fit concatenations (tests/test_models.py:83-147: LinearBasis + RandomRBF + RandomMatern52):
    every child with its own regulariser over its column slice (basis_functions.py:1712-1748), its own length scales (ARD and
    isotropic here), its own columns of X (apply_ind); the EdPhi product is stored and contracted child by child."""
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    from revrand_amd import _hip
    X, y, largs = _data(lik)
    d = X.shape[1]
    steps = [0]
    real = _hip.ResidentSgd.step

    def spy(self, *a, **k):
        steps[0] += 1
        return real(self, *a, **k)
    monkeypatch.setattr(_hip.ResidentSgd, "step", spy)
    like = {"poisson": lk.Poisson, "gaussian": lk.Gaussian, "binomial": lk.Binomial}[lik]
    out = []
    for resident in (True, False):
        basis = bs.RandomRBF(nbases=48, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()),
                             regularizer=Parameter(1.5, Positive())) \
            + bs.LinearBasis(onescol=True, regularizer=Parameter(0.7, Positive())) \
            + bs.RandomMatern52(nbases=24, Xdim=2, random_state=2, apply_ind=[0, 2], lenscale=Parameter(0.8, Positive()))
        glm = GLM(like(), basis, K=3, nsamples=8, batch_size=1500, maxiter=20, nstarts=2, random_state=11)
        glm._resident_sgd = resident
        np.random.seed(3)
        steps[0] = 0
        glm.fit(X, y, likelihood_args=largs)
        assert steps[0] == (20 if resident else 0)

        def flat(v):
            v = v if isinstance(v, (list, tuple)) else [v]
            return np.concatenate([np.atleast_1d(np.asarray(u, dtype=float)).ravel() for u in v] + [np.empty(0)])
        out.append((glm.weights_.copy(), glm.covariance_.copy(), flat(glm.regularizer_), flat(glm.like_hypers_),
                    flat([h for h in glm.basis_hypers_ if np.size(h)]), glm.random_.randn()))
        assert out[-1][2].shape == (3,) and out[-1][4].shape == (d + 1,)
    _same(out[0], out[1], 2e-5)


def test_linear_basis_alone():
    """No length scale at all: the loop runs without the length-scale half of its update."""
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    X, y, _ = _data("gaussian")
    out = []
    for resident in (True, False):
        glm = GLM(lk.Gaussian(), bs.LinearBasis(onescol=True), K=2, nsamples=6, batch_size=800, maxiter=15, nstarts=0, random_state=4)
        glm._resident_sgd = resident
        np.random.seed(2)
        glm.fit(X, y)
        out.append((glm.weights_.copy(), glm.covariance_.copy(), np.atleast_1d(float(glm.regularizer_)),
                    np.atleast_1d(np.asarray(glm.like_hypers_, dtype=float)), np.empty(0), glm.random_.randn()))
    _same(out[0], out[1], 2e-5)


def test_fits_the_loop_does_not_cover_take_the_host_loop(monkeypatch):
    """A custom updater, K > 64: `_resident_loop` declines and `fit` is what it was."""
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    from revrand_amd import _hip
    monkeypatch.setattr(_hip.ResidentSgd, "step", lambda *a, **k: (_ for _ in ()).throw(AssertionError("resident loop used")))
    monkeypatch.setattr(_hip.FusedSvi, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("fused loop used")))
    X, y, _ = _data("poisson", N=1200)
    d = X.shape[1]

    class MyAdam(opt.Adam):
        pass
    for basis, kw in ((bs.RandomRBF(nbases=16, Xdim=d, random_state=1), {"updater": MyAdam()}),
                      (bs.RandomRBF(nbases=16, Xdim=d, random_state=1), {"K": 65})):
        glm = GLM(lk.Poisson(), basis, nsamples=4, batch_size=300, maxiter=3, nstarts=0, random_state=1, **{"K": 2, **kw})
        glm.fit(X, y)
        assert np.all(np.isfinite(glm.weights_))


@pytest.mark.parametrize("batch", [1500, 12])
def test_fastfood_children_and_many_components_run_resident(batch, monkeypatch):
    """Round 6: a FastFoodRBF child takes the resident loops through its dense equivalent W = _makeVX(I_d) (the chain is linear in
    x: basis_functions.py:1263-1289, 1356-1371) -- alone (ARD) and in a concatenation -- and K up to 64 components
    (rr_glm_sgd) / 32 (the fused small-batch loop): the host loop's fit, which runs the CHAIN kernel, from the same seeds."""
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    from revrand_amd import _hip
    X, y, _ = _data("poisson", N=3000)
    d = X.shape[1]
    steps = {"resident": 0, "fused": 0}
    real, real_run = _hip.ResidentSgd.step, _hip.FusedSvi.run

    def spy(self, *a, **k):
        steps["resident"] += 1
        return real(self, *a, **k)

    def spy_run(self, n, *a, **k):
        steps["fused"] += n
        return real_run(self, n, *a, **k)
    monkeypatch.setattr(_hip.ResidentSgd, "step", spy)
    monkeypatch.setattr(_hip.FusedSvi, "run", spy_run)

    def bases():
        return [bs.FastFoodRBF(nbases=24, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                bs.LinearBasis(onescol=True) + bs.FastFoodRBF(nbases=16, Xdim=d, random_state=2)
                + bs.RandomRBF(nbases=8, Xdim=d, random_state=3)]
    for which in (0, 1):
        for K in ((3, 40) if batch > 100 else (3, 20)):
            out = []
            for resident in (True, False):
                glm = GLM(lk.Poisson(), bases()[which], K=K, nsamples=6, batch_size=batch, maxiter=12, nstarts=2, random_state=5)
                glm._resident_sgd = resident
                np.random.seed(3)
                steps["resident"] = steps["fused"] = 0
                glm.fit(X, y)
                if resident:
                    assert steps["resident" if batch > 100 else "fused"] == 12, (steps, which, K)
                else:
                    assert steps == {"resident": 0, "fused": 0}
                out.append((glm.weights_.copy(), glm.covariance_.copy(), np.atleast_1d(np.array(glm.regularizer_, dtype=float)),
                            np.zeros(0), np.concatenate([np.ravel(np.asarray(h, dtype=float)) for h in
                                                         (glm.basis_hypers_ if isinstance(glm.basis_hypers_, list) else [glm.basis_hypers_])]),
                            glm.random_.randn()))
            _same(out[0], out[1], 5e-5)


def _flat(v):
    if isinstance(v, (list, tuple)):
        return np.concatenate([_flat(u) for u in v]) if len(v) else np.zeros(0)
    return np.atleast_1d(np.asarray(v, dtype=float)).ravel()


@pytest.mark.parametrize("lik", ["poisson", "gaussian"])
@pytest.mark.parametrize("batch", [1500, 12])
def test_spectral_mixture_children_run_resident(lik, batch, monkeypatch):
    """A FastFoodGM child (basis_functions.py:1386-1562: [cos | sin](VX + mX) | [cos | sin](VX - mX), parameters mean AND length
    scales) -- alone and in `Linear + FastFoodGM + RandomRBF` -- on the step-per-call loop: its two blocks are random Fourier
    blocks of the chain's dense equivalent with every frequency moved by +- mean (RR_SGD_CHILD_GM), its two gradients sums over
    their contractions.  Against the host loop, whose features come from the CHAIN kernel's mixture mode; small minibatches too
    (the fused many-steps-per-launch kernel does not take this child: the step-per-call loop does)."""
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    from revrand_amd import _hip
    X, y, _ = _data(lik, N=3000, d=12)
    d = X.shape[1]
    steps = [0]
    real = _hip.ResidentSgd.step

    def spy(self, *a, **k):
        steps[0] += 1
        return real(self, *a, **k)
    monkeypatch.setattr(_hip.ResidentSgd, "step", spy)
    monkeypatch.setattr(_hip.FusedSvi, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("fused loop used")))
    like = {"poisson": lk.Poisson, "gaussian": lk.Gaussian}[lik]

    def bases():
        return [bs.FastFoodGM(nbases=24, Xdim=d, random_state=1),
                bs.LinearBasis(onescol=True) + bs.FastFoodGM(nbases=16, Xdim=d, random_state=2, mean=Parameter(0.3 * np.ones(d), Bound()),
                                                              lenscale=Parameter(1.2 * np.ones(d), Positive()))
                + bs.RandomRBF(nbases=8, Xdim=d, random_state=3)]
    for which in (0, 1):
        out = []
        for resident in (True, False):
            glm = GLM(like(), bases()[which], K=3, nsamples=6, batch_size=batch, maxiter=12, nstarts=2, random_state=5)
            glm._resident_sgd = resident
            np.random.seed(3)
            steps[0] = 0
            glm.fit(X, y)
            assert steps[0] == (12 if resident else 0), (steps, which)
            out.append((glm.weights_.copy(), glm.covariance_.copy(), _flat(glm.regularizer_), _flat(glm.like_hypers_),
                        _flat(glm.basis_hypers_), glm.random_.randn()))
        assert out[0][4].size == (2 * d if which == 0 else 2 * d + 1)
        _same(out[0], out[1], 1e-4)


def test_config5_shape_runs_the_fused_contraction_and_improves_the_objective(monkeypatch):
    """F = 2048, D = 32 ARD, K L = 500, minibatch 16 384 (config 5 with a quarter of its rows per step): the plan that contracts
    EdPhi in registers is taken inside the resident loop; 30 steps from the same start give the host loop's parameters."""
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    monkeypatch.delenv("RR_GLM_SGD_OVERLAP", raising=False)   # the library's own decision at this size: two streams
    rs = np.random.RandomState(0)
    N, d, n = 60000, 32, 1024
    X = rs.randn(N, d).astype(np.float32)
    y = rs.poisson(np.exp(0.3 * X[:, 0].astype(np.float64))).astype(float)
    out = []
    for resident in (True, False):
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
        glm = GLM(lk.Poisson(), basis, K=10, nsamples=50, batch_size=16384, maxiter=30, nstarts=0, random_state=2)
        glm._resident_sgd = resident
        np.random.seed(5)
        glm.fit(X, y)
        out.append((glm.weights_.copy(), glm.covariance_.copy(), np.array(glm.basis_hypers_), float(glm.regularizer_)))
    (wa, ca, ha, ra), (wb, cb, hb, rb) = out
    assert normwise(wa, wb) < 1e-4 and normwise(ca, cb) < 1e-4 and normwise(ha, hb) < 1e-4 and abs(ra - rb) < 1e-4 * rb


def test_edge_shapes_refits_clone_and_pickle():
    """The loop at the edges: one mixture component, one step, a batch larger than the data (batch = N), an estimator fitted
    twice, cloned and pickled afterwards, then serving -- each against the host loop."""
    import pickle
    from sklearn.base import clone
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    X, y, _ = _data("poisson", N=700)
    d = X.shape[1]

    def make(resident, **kw):
        basis = bs.RandomRBF(nbases=24, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
        g = GLM(lk.Poisson(), basis, nsamples=5, random_state=7, nstarts=0, **kw)
        g._resident_sgd = resident
        return g

    def fitted(g):
        np.random.seed(4)
        g.fit(X, y)
        return np.concatenate((g.weights_.ravel(), g.covariance_.ravel(), np.atleast_1d(g.basis_hypers_), [float(g.regularizer_)]))
    for kw in ({"K": 1, "batch_size": 64, "maxiter": 9}, {"K": 2, "batch_size": 64, "maxiter": 1},
               {"K": 2, "batch_size": 5000, "maxiter": 6}, {"K": 3, "batch_size": 10, "maxiter": 40}):
        a, b = fitted(make(True, **kw)), fitted(make(False, **kw))
        assert normwise(a, b) < 2e-5, (kw, normwise(a, b))
    g = make(True, K=2, batch_size=100, maxiter=8)
    first = fitted(g)
    g.random_ = np.random.RandomState(7)
    assert normwise(fitted(g), first) < 2e-5          # a second fit of the same estimator starts from scratch
    assert np.all(np.isfinite(g.predict(X[:50])))      # serving after a resident fit
    g2 = pickle.loads(pickle.dumps(g))
    assert np.array_equal(g2.weights_, g.weights_) and "_resident_clock" not in g2.__dict__
    c = clone(g)
    assert c.get_params()["K"] == 2 and not hasattr(c, "weights_")
