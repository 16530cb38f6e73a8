"""CPU tests of the host-side mirror: parameter types, W sampling against the golden vectors,
concatenation bookkeeping, optimiser front-ends.  Modelled on the reference's
tests/test_bases.py, test_btypes.py and test_optimize.py (no device needed)."""
import os
import pickle

import numpy as np
import pytest
from scipy.optimize import minimize
from scipy.stats import gamma, norm

import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Bound, Parameter, Positive
from revrand_amd.linalg import solve_posdef
from revrand_amd.optimize import logtrick_minimizer, structured_minimizer
from revrand_amd.utils import flatten_values, unflatten

import revrand_oracle as orc  # the checker (tests only)

CLASSES = ["RandomRBF", "RandomLaplace", "RandomCauchy", "RandomMatern32", "RandomMatern52", "OrthogonalRBF"]


@pytest.mark.parametrize("cname", CLASSES)
def test_weight_sampling_matches_reference(golden, cname):
    g = golden("weights")
    for (d, n, seed) in [(3, 4, 7), (8, 256, 1)]:
        b = getattr(bs, cname)(nbases=n, Xdim=d, random_state=seed)
        assert np.array_equal(b.W, g["%s_d%d_n%d_s%d" % (cname, d, n, seed)])


def test_bounds_and_parameters():
    assert Bound(1, 2).check(1.5) and not Bound(1, 2).check(3)
    assert Bound(1, 2).clip(3) == 2
    with pytest.raises(ValueError):
        Bound(42, 10)
    assert Positive().lower == 1e-14 and repr(Positive()) == "Positive(upper=None)"
    p = Parameter()
    assert p.value == [] and not p.has_value
    p = Parameter(1.2, Bound(1, 2))
    assert p.shape == () and p.rvs() == 1.2 and not p.is_random and p.is_scalar
    p = Parameter(gamma(a=1, scale=1), Positive(), shape=(3,))
    assert np.all(p.value == np.ones(3)) and p.rvs(0).shape == (3,) and p.is_random
    with pytest.raises(ValueError):
        Parameter(-1., Positive())
    assert pickle.loads(pickle.dumps(Positive(3.))) == Positive(3.)


def test_concat_params_and_regularizer():
    d, D = 10, 20
    base = bs.LinearBasis(onescol=True) + bs.RandomMatern52(nbases=D, Xdim=d, lenscale=Parameter(1., Positive()))
    assert np.isscalar(base.params.value)
    base = bs.LinearBasis(onescol=True) + bs.RandomMatern52(nbases=D, Xdim=d,
                                                            lenscale=Parameter(np.ones(d), Positive()))
    assert len(base.params.value) == d
    base += bs.RandomMatern52(nbases=D, Xdim=d, lenscale=Parameter(1., Positive()))
    assert len(base.params) == 2
    X = np.random.RandomState(0).randn(7, d)
    diag, slices = base.regularizer_diagonal(X, 2.0, 3.0, 4.0)
    assert slices == [slice(0, d + 1), slice(d + 1, d + 1 + 2 * D), slice(d + 1 + 2 * D, d + 1 + 4 * D)]
    assert np.all(diag[:d + 1] == 2.0) and np.all(diag[-2 * D:] == 4.0)
    assert base.get_dim(X) == d + 1 + 4 * D
    # single basis
    b = bs.LinearBasis(regularizer=Parameter(2, Positive()))
    dg, sl = b.regularizer_diagonal(X)
    assert np.all(dg == 2.0) and sl == slice(None) and len(dg) == d + 1
    with pytest.raises(ValueError, match="scalar"):
        bs.LinearBasis(regularizer=Parameter(np.ones(2), Positive()))
    with pytest.raises(ValueError, match="bounded below"):
        bs.LinearBasis(regularizer=Parameter(1., Bound(-1., None)))


def test_linear_concat_transform_and_empty_grads():
    X = np.random.RandomState(1).randn(9, 3)
    base = bs.LinearBasis(onescol=False) + bs.LinearBasis(onescol=False)
    assert np.allclose(base.transform(X), np.hstack((X, X)))
    assert list(base.grad(X)) == []
    assert sum([bs.LinearBasis(), bs.BiasBasis()]).get_dim(X) == 5
    sl = bs.LinearBasis(onescol=False, apply_ind=[0]) + bs.LinearBasis(onescol=True, apply_ind=slice(1, 3))
    assert np.allclose(sl.transform(X), np.hstack((X[:, [0]], np.ones((9, 1)), X[:, 1:3])))


def test_lenscale_validation_messages():
    with pytest.raises(ValueError, match="Parameter dimension doesn't agree"):
        bs.RandomRBF(nbases=4, Xdim=3, lenscale=Parameter(np.ones(2), Positive()))
    b = bs.RandomRBF(nbases=4, Xdim=3, random_state=0)
    with pytest.raises(ValueError, match="Dimensions of data inconsistent!"):
        b._check_dim(4, None)
    with pytest.raises(ValueError, match="Dimension of input parameter is inconsistent!"):
        b._check_dim(3, np.ones(3))
    assert b._check_dim(3, None)[0] == 1.0
    assert bs.count_args(b.transform) == 2 and bs.count_args(b.grad) == 2
    assert "RandomRBF(nbases=4, Xdim=3" in repr(b)


def test_basis_pickles_without_device_handle():
    b = bs.RandomRBF(nbases=4, Xdim=3, random_state=0)
    b.__dict__["_hip_handle"] = (0, object())
    b2 = pickle.loads(pickle.dumps(b))
    assert "_hip_handle" not in b2.__dict__ and np.array_equal(b2.W, b.W)


def test_apply_grad_structures():
    f = lambda g: g.sum()  # noqa: E731
    assert bs.apply_grad(f, []) == []
    assert bs.apply_grad(f, np.ones((3, 4))) == 12
    assert bs.apply_grad(f, np.ones((3, 4, 5))).shape == (5,)
    out = bs.apply_grad(f, iter([np.ones((2, 2)), np.ones((2, 2, 3))]))
    assert out[0] == 4 and out[1].shape == (3,)
    assert bs.apply_grad(f, [np.ones((2, 2))]) == 4
    with pytest.raises(ValueError):
        bs.apply_grad(f, np.ones((1, 1, 1, 1)))


def test_flatten_roundtrip():
    nested = [1.5, [2.0, np.arange(3.)], [], np.ones((2, 2))]
    shapes = [(), [(), (3,)], (0,), (2, 2)]
    flat = flatten_values(nested)
    assert flat.shape == (9,)
    back = unflatten(flat, shapes)
    assert back[0] == 1.5 and back[2] == [] and np.array_equal(back[1][1], np.arange(3.))
    assert back[3].shape == (2, 2)


def test_structured_logtrick_minimizer():
    """Quadratic fit as in the reference's tests/test_optimize.py (tolerance 1e-3)."""
    rs = np.random.RandomState(99)
    x = np.linspace(-1, 1, 1000)
    a, b, c = 3., 2., 1.
    yv = a * x ** 2 + b * x + c + rs.randn(1000) * 1e-4
    A = np.vstack((x ** 2, x, np.ones_like(x))).T

    def obj(w, cc):
        r = A[:, :2] @ w + cc - yv
        return 0.5 * (r ** 2).sum(), [A[:, :2].T @ r, r.sum()]

    nmin = structured_minimizer(logtrick_minimizer(minimize))
    res = nmin(obj, [Parameter(np.array([1., 1.]), Positive()), Parameter(0.5, Bound(0.1, None))],
               method="L-BFGS-B", jac=True)
    w, cc = res.x
    assert np.allclose(w, [a, b], atol=1e-3) and abs(cc - c) < 1e-3
    # random starts pick the best candidate and keep the structure
    res = nmin(obj, [Parameter(gamma(2.), Positive(), shape=(2,)), Parameter(norm(1., .1), Bound())],
               method="L-BFGS-B", jac=True, nstarts=20, random_state=np.random.RandomState(1))
    assert np.allclose(res.x[0], [a, b], atol=1e-3) and np.isscalar(res.x[1])


def test_solve_posdef(golden):
    g = golden("solve_posdef")
    for tag in ("pd", "npd"):
        X, ld = solve_posdef(g[tag + "_A"], np.eye(5))
        assert np.abs(X - g[tag + "_X"]).max() < 1e-9 * np.abs(g[tag + "_X"]).max()
        if np.isfinite(g[tag + "_logdet"]):
            assert abs(ld - g[tag + "_logdet"]) < 1e-9 * abs(ld)


@pytest.mark.parametrize("case", [(1, 10), (2, 10), (5, 16), (16, 64), (128, 256)])
def test_fastfood_matrices_match_reference(golden, case):
    d, nb = case
    g = golden("fastfood")
    k = "d%d_nb%d" % (d, nb)
    b = bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=3)
    assert np.array_equal(b.B, g[k + "_B"]) and np.array_equal(b.PI, g[k + "_PI"])
    assert np.array_equal(b.G, g[k + "_G"])
    assert np.abs(b.S - g[k + "_S"]).max() < 1e-13 * np.abs(b.S).max()
    assert b.n == b.d2 * b.k and b.get_dim(None) == 2 * b.n
    assert "FastFoodRBF(nbases=%d, Xdim=%d" % (nb, d) in repr(b)


# -- GLM host side: updaters, SGD front-ends, likelihoods, mixture terms ---------------------------

def test_sgd_updaters_match_reference(golden):
    from revrand_amd import optimize as opt
    g = golden("glm")
    for name, upd in (("sgd", opt.SGDUpdater()), ("adadelta", opt.AdaDelta()), ("adagrad", opt.AdaGrad()),
                      ("momentum", opt.Momentum()), ("adam", opt.Adam())):
        for _ in range(2):   # reset() must restore the initial state
            upd.reset()
            x = np.linspace(-1, 1, 6)
            for i, grad in enumerate(g["upd_grads"]):
                x = upd(x, grad)
                assert np.allclose(x, g["upd_" + name][i], rtol=1e-12, atol=1e-14), name
    with pytest.raises(ValueError):
        opt.AdaDelta(rho=2)
    with pytest.raises(ValueError):
        opt.AdaGrad(eta=0)
    with pytest.raises(ValueError):
        opt.Momentum(rho=-1)


def test_sgd_and_structured_logtrick_sgd_fit_a_line():
    """Least squares by minibatch SGD, plain and through the structured + log-trick front-ends
    (mirrors the reference's tests/test_optimize.py use of sgd)."""
    from revrand_amd import optimize as opt
    rs = np.random.RandomState(0)
    N = 400
    x = np.linspace(-1, 1, N)
    Xd = np.column_stack((np.ones(N), x))
    y = Xd @ np.array([0.5, 2.0]) + 0.01 * rs.randn(N)

    def grad(w, Xb, yb):
        return -2 * Xb.T @ (yb - Xb @ w) / len(yb)

    res = opt.sgd(grad, np.zeros(2), data=(Xd, y), maxiter=2000, batch_size=20, random_state=1,
                  updater=opt.AdaGrad())
    assert np.allclose(res.x, [0.5, 2.0], atol=0.05) and res.fun is None and len(res.norms) == 2000

    def obj(w, s, Xb, yb):   # s > 0 scales nothing; its gradient is 0 -> stays put under the log trick
        r = yb - Xb @ w
        return (r ** 2).mean(), [-2 * Xb.T @ r / len(yb), 0.0]

    nsgd = opt.structured_sgd(opt.logtrick_sgd(opt.sgd))
    res = nsgd(obj, [Parameter(norm(), Bound(), shape=(2,)), Parameter(1.5, Positive())], (Xd, y), eval_obj=True,
               maxiter=2000, batch_size=20, random_state=rs, nstarts=20, updater=opt.AdaGrad())
    w, s = res.x
    assert np.allclose(w, [0.5, 2.0], atol=0.05) and abs(s - 1.5) < 1e-12
    assert len(res.objs) == 2000
    # bounds are honoured
    res = opt.sgd(grad, np.zeros(2), data=(Xd, y), maxiter=500, batch_size=20, random_state=1,
                  bounds=[Bound(None, 0.2), Bound(0, None)], updater=opt.AdaGrad())
    assert res.x[0] <= 0.2 + 1e-12
    with pytest.raises(ValueError):
        opt.sgd(grad, np.zeros(2), data=(Xd, y), bounds=[Bound()])


def test_gen_batch_sweeps_permutations():
    from revrand_amd.optimize import gen_batch
    data = np.arange(10)
    seen = np.concatenate([b[0] for b in gen_batch(data, 5, maxiter=2, random_state=0)])
    assert sorted(seen) == list(range(10))
    b = next(gen_batch((data, data * 2), 3, random_state=0))
    assert np.array_equal(b[1], 2 * b[0])


def test_likelihoods_match_oracle_formulas():
    import revrand_oracle as orc
    from revrand_amd import likelihoods as lk
    rs = np.random.RandomState(0)
    f = 2 * rs.randn(3, 50)
    n = rs.randint(1, 9, 50).astype(float)
    cases = [("bernoulli", lk.Bernoulli(), (rs.rand(50) < 0.5).astype(float), ()),
             ("binomial", lk.Binomial(), np.minimum(rs.poisson(2, 50), n).astype(float), (n,)),
             ("gaussian", lk.Gaussian(), rs.randn(50), (0.7,)),
             ("poisson_exp", lk.Poisson("exp"), rs.poisson(2, 50).astype(float), ()),
             ("poisson_softplus", lk.Poisson("softplus"), rs.poisson(2, 50).astype(float), ())]
    for name, L, y, args in cases:
        assert np.allclose(L.loglike(y, f, *args), orc.lik_loglike(name, y, f, *args), rtol=1e-10, atol=1e-12), name
        assert np.allclose(L.df(y, f, *args), orc.lik_df(name, y, f, *args), rtol=1e-10, atol=1e-12), name
        dp = L.dp(y, f, *args)
        if name == "gaussian":
            assert np.allclose(dp, orc.lik_dp(name, y, f, *args)[0])
        else:
            assert dp == []
        lid, par, rowarg, const = L.device_spec(y, list(args) if name == "gaussian" else [], args)
        assert (rowarg is not None) == (name == "binomial") and np.isfinite(const)
        assert np.all(np.isfinite(L.Ey(f, *args))) and np.all((L.cdf(y, f, *args) >= 0) & (L.cdf(y, f, *args) <= 1))
    with pytest.raises(ValueError):
        lk.Poisson("log")
    with pytest.raises(ValueError):
        lk.Gaussian().loglike(0., 0., -1.0)


def test_glm_qmatrix_and_clone():
    import revrand_oracle as orc
    from sklearn.base import clone
    from revrand_amd import glm as G
    from revrand_amd.likelihoods import Gaussian
    rs = np.random.RandomState(1)
    m, C = rs.randn(7, 4), rs.gamma(2., 0.5, (7, 4))
    assert np.allclose(G._qmatrix(m, C), orc.glm_qmatrix(m, C), rtol=1e-12)
    glm = G.GeneralizedLinearModel(likelihood=Gaussian(), basis=bs.LinearBasis(onescol=True), K=3, maxiter=10,
                                   batch_size=5, nsamples=4, nstarts=2, random_state=0)
    c = clone(glm)
    for k in ("K", "maxiter", "batch_size", "nsamples", "nstarts", "random_state"):
        assert glm.get_params()[k] == c.get_params()[k]
    assert repr(c.basis) == repr(glm.basis) and isinstance(G.GeneralisedLinearModel(), G.GeneralizedLinearModel)
    assert G._reshape_likelihood_args((2., np.arange(3.)), 3)[0].shape == (3,)
    with pytest.raises(ValueError):
        G._reshape_likelihood_args((np.arange(4.),), 3)


def test_gen_batch_is_the_endless_permutation_stream():
    """The vectorised minibatch generator consumes the RandomState exactly like the reference's per-index
    generator (utils/rand.py:7-31), including draws the caller makes in between (the GLM's randn)."""
    from revrand_amd.optimize import gen_batch
    from revrand_amd.utils import endless_permutations
    N, B = 23, 7
    rs1, rs2 = np.random.RandomState(3), np.random.RandomState(3)
    g = gen_batch(np.arange(N), B, maxiter=12, random_state=rs1)
    p = endless_permutations(N, rs2)
    for _ in range(12):
        a = next(g)[0]
        b = np.array([next(p) for _ in range(B)])
        assert np.array_equal(a, b)
        assert rs1.randn() == rs2.randn()


def test_likelihoods_like_reference_test_likelihoods():
    """tests/test_likelihoods.py of the reference: shapes of every method, and the densities / CDFs against
    scipy.stats (with f on the link scale)."""
    from scipy.stats import bernoulli, binom, poisson
    from scipy.special import logit
    from revrand_amd import likelihoods as lk
    N = 100
    y, f = np.ones(N), np.ones(N) * 2
    for like, args in zip([lk.Gaussian, lk.Poisson, lk.Bernoulli, lk.Binomial], [[1.], [], [], [5]]):
        lobj = like()
        for out in (lobj.loglike(y, f, *args), lobj.Ey(f, *args), lobj.df(y, f, *args), lobj.cdf(y, f, *args)):
            assert np.shape(out) == (N,)
        dp = lobj.dp(y, f, *args)
        assert (np.shape(dp) == (N,)) if like is lk.Gaussian else (dp == [])
    x = np.linspace(-10, 10, 100)
    assert np.allclose(lk.Gaussian().loglike(x, 0., 2.), norm.logpdf(x, loc=0, scale=np.sqrt(2)))
    assert np.allclose(lk.Gaussian().cdf(x, 0., 2.), norm.cdf(x, loc=0, scale=np.sqrt(2)))
    xb = np.array([0, 1])
    assert np.allclose(lk.Bernoulli().loglike(xb, logit(0.3)), bernoulli.logpmf(xb, 0.3))
    assert np.allclose(lk.Bernoulli().cdf(xb, logit(0.3)), bernoulli.cdf(xb, 0.3))
    xn = np.arange(6)
    assert np.allclose(lk.Binomial().loglike(xn, logit(0.3), 5), binom.logpmf(xn, p=0.3, n=5))
    assert np.allclose(lk.Binomial().cdf(xn, logit(0.3), 5), binom.cdf(xn, p=0.3, n=5))
    assert np.allclose(lk.Poisson().loglike(xn, np.log(2.)), poisson.logpmf(xn, 2.))
    assert np.allclose(lk.Poisson().cdf(xn, np.log(2.)), poisson.cdf(xn, 2.))
    g = np.log(np.expm1(2.))   # softplus(g) = 2
    assert np.allclose(lk.Poisson("softplus").loglike(xn, g), poisson.logpmf(xn, 2.))


def test_sgd_prefetch_changes_nothing_but_the_thread():
    """`prefetch=True` builds each minibatch one step ahead on a worker thread: same batches, same result, the
    generator's exceptions still surface, and an abandoned run does not leave the worker blocked."""
    import threading
    from revrand_amd.optimize import sgd, Adam, _prefetched
    rs = np.random.RandomState(0)
    X = rs.randn(500, 3)
    w = np.array([1.0, -2.0, 0.5])
    y = X @ w

    def grad(wv, Xb, yb):
        return 2 * Xb.T @ (Xb @ wv - yb) / len(yb)

    res = [sgd(grad, np.zeros(3), [X, y], batch_size=37, maxiter=200, updater=Adam(alpha=0.05),
               random_state=np.random.RandomState(5), prefetch=pf) for pf in (False, True)]
    assert np.array_equal(res[0].x, res[1].x) and res[0].norms == res[1].norms
    assert np.allclose(res[1].x, w, atol=1e-2)

    def boom():
        yield 1
        raise RuntimeError("from the generator")
    it = _prefetched(boom())
    assert next(it) == 1
    with pytest.raises(RuntimeError, match="from the generator"):
        next(it)
    before = threading.active_count()
    it = _prefetched(iter(range(1000)))
    assert next(it) == 0
    it.close()                      # consumer walks away: the worker must notice and end
    import time
    for _ in range(50):
        if threading.active_count() <= before:
            break
        time.sleep(0.05)
    assert threading.active_count() <= before


def test_prefetch_pipeline_of_several_stages_keeps_order_and_surfaces_errors():
    """A list of callables is a pipeline -- the first on the thread that cuts the batches, each further one on its own thread:
    every batch passes every stage once, in order; a stage's exception reaches the consumer; an abandoned run ends all workers;
    `sgd` takes the list as `prefetch` and its device-loop protocol (begin / step / end) sees the augmented batches."""
    import threading
    import time
    from revrand_amd.optimize import sgd, _prefetched
    seen = [[], []]

    def a(x):
        seen[0].append((threading.get_ident(), x))
        return x * 2

    def b(x):
        time.sleep(0.001)
        seen[1].append((threading.get_ident(), x))
        return x + 1
    out = list(_prefetched(iter(range(50)), [a, b]))
    assert out == [2 * i + 1 for i in range(50)]
    assert [x for _, x in seen[0]] == list(range(50)) and [x for _, x in seen[1]] == [2 * i for i in range(50)]
    assert len({t for t, _ in seen[0]}) == 1 and len({t for t, _ in seen[1]}) == 1 and seen[0][0][0] != seen[1][0][0]
    assert threading.get_ident() not in {seen[0][0][0], seen[1][0][0]}

    def bad(x):
        if x == 6:
            raise ValueError("stage two")
        return x
    it = _prefetched(iter(range(50)), [a, bad])
    assert [next(it) for _ in range(3)] == [0, 2, 4]
    with pytest.raises(ValueError, match="stage two"):
        next(it)
    before = threading.active_count()
    it = _prefetched(iter(range(10 ** 6)), [a, b, b])
    assert next(it) == 2
    it.close()
    for _ in range(50):
        if threading.active_count() <= before:
            break
        time.sleep(0.05)
    assert threading.active_count() <= before

    class Loop(object):  # the protocol of glm._ResidentLoop
        def begin(self, x0, lower, upper, updater, maxiter):
            self.x, self.batches, self.maxiter = np.array(x0), [], maxiter
            assert np.all(lower == -1.0) and np.all(upper == np.inf)

        def step(self, batch):
            self.batches.append(batch)
            self.x = self.x + 1

        def end(self):
            return self.x, np.arange(len(self.batches), dtype=float), np.ones(len(self.batches))

        def abort(self):
            self.aborted = True
    X = np.arange(40.0).reshape(20, 2)
    loop = Loop()
    res = sgd(None, np.zeros(2), [X], bounds=[(-1.0, None)] * 2, batch_size=5, maxiter=7, random_state=np.random.RandomState(1),
              prefetch=[lambda bt: list(bt) + ["drawn"], lambda bt: list(bt) + ["uploaded"]], args=("arg",), device_loop=loop)
    assert np.array_equal(res.x, [7.0, 7.0]) and res.objs == list(range(7)) and res.fun == 6.0 and len(res.norms) == 7
    assert all(bt[1:] == ["drawn", "uploaded", "arg"] and bt[0].shape == (5, 2) for bt in loop.batches)

    class Failing(Loop):
        def step(self, batch):
            raise RuntimeError("device step")
    bad_loop = Failing()
    with pytest.raises(RuntimeError, match="device step"):
        sgd(None, np.zeros(2), [X], bounds=[(-1.0, None)] * 2, batch_size=5, maxiter=7, prefetch=True, device_loop=bad_loop)
    assert bad_loop.aborted


def test_poisson_log_factorial_table():
    from scipy.special import gammaln
    from revrand_amd.likelihoods import _sum_gammaln1p
    rs = np.random.RandomState(1)
    for y in (rs.poisson(3.0, size=1000).astype(float), np.zeros(5), np.array([]), rs.rand(20) * 7, np.array([5000.0, 2.0])):
        assert abs(_sum_gammaln1p(y) - float(gammaln(y + 1).sum())) <= 1e-12 * max(1.0, abs(float(gammaln(y + 1).sum())))


def test_gram_engine_keyword_is_a_plain_sklearn_parameter():
    """No device needed: the keyword is stored verbatim, survives clone / get_params / set_params and pickling."""
    import pickle
    from sklearn.base import clone
    import revrand_amd.basis_functions as bs
    from revrand_amd import StandardLinearModel, GeneralizedLinearModel
    slm = StandardLinearModel(bs.LinearBasis(onescol=True), gram_engine="fp16x3")
    assert slm.get_params()["gram_engine"] == "fp16x3" and clone(slm).gram_engine == "fp16x3"
    assert StandardLinearModel().gram_engine is None
    assert pickle.loads(pickle.dumps(slm)).gram_engine == "fp16x3"
    glm = GeneralizedLinearModel(gram_engine="bf16x3")
    assert clone(glm).set_params(gram_engine=None).gram_engine is None


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_library_generator_reproduces_numpy_legacy_randn(dtype):
    """rr_legacy_randn (MT19937 words + polar method on the calling thread, sqrt / log on worker threads) returns what
    `RandomState.randn` returns, bit for bit, and leaves the same generator state: odd and even counts (the cached second
    value of a pair), counts that cross the 624-word regeneration and the worker hand-over blocks, interleaved with other
    draws of the same RandomState as the SVI loop interleaves minibatch permutations."""
    from revrand_amd import _hip
    for seed, sizes in [(0, [1, 2, 3, 7, 311, 312, 313, 1000, 65537, 200001]), (123, [5, 5, 4, 100000, 1, 131072 + 3])]:
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        for i, n in enumerate(sizes):
            want = a.randn(n).astype(dtype)
            got = _hip.legacy_randn(b, n, dtype, threads=4)
            assert got.dtype == np.dtype(dtype) and np.array_equal(got, want), (seed, n)
            if i % 2:  # other consumers of the stream in between
                assert np.array_equal(a.permutation(17), b.permutation(17))
                assert a.randint(0, 1000) == b.randint(0, 1000)
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
    # the GLM's use: K calls of randn(L, D) == one call of K L D values
    a, b = np.random.RandomState(9), np.random.RandomState(9)
    want = np.concatenate([a.randn(5, 33) for _ in range(3)]).astype(dtype)
    assert np.array_equal(_hip.legacy_randn(b, 3 * 5 * 33, dtype).reshape(15, 33), want)
    assert a.randn() == b.randn()


def test_library_generator_reproduces_numpy_legacy_permutation():
    """rr_legacy_permutation returns what `RandomState.permutation(n)` returns, bit for bit, and leaves the same generator
    state (position, cached Gaussian): lengths around powers of two (the mask of random_interval changes there), around the
    block the partners are drawn in, interleaved with randn as the SVI loop interleaves them; and the minibatch stream of
    `gen_batch` built on it is the stream built on NumPy's own."""
    from revrand_amd import _hip
    from revrand_amd import optimize as opt
    for seed in (0, 77):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        for n in (4096, 4097, 5000, 8191, 8192, 8193, 65537, 300001):
            assert a.randn(3).tolist() == b.randn(3).tolist()   # an odd count: a cached Gaussian must survive the call
            want, got = a.permutation(n), _hip.legacy_permutation(b, n)
            assert got.dtype == want.dtype and np.array_equal(got, want), (seed, n)
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
    assert np.array_equal(_hip.legacy_permutation(np.random.RandomState(3), 100), np.random.RandomState(3).permutation(100))  # short: NumPy's
    N, M = 10000, 3000   # batches that straddle epoch boundaries
    ra, rb = np.random.RandomState(5), np.random.RandomState(5)
    data = [np.arange(N)]
    ref_perm, pos, want = np.empty(0, dtype=int), 0, []
    for _ in range(12):
        parts, need = [], M
        while need:
            if pos == len(ref_perm):
                ref_perm, pos = ra.permutation(N), 0
            take = min(need, len(ref_perm) - pos)
            parts.append(ref_perm[pos:pos + take]); pos += take; need -= take
        want.append(np.concatenate(parts))
    got = [bt[0] for bt in opt.gen_batch(data, M, 12, rb)]
    assert all(np.array_equal(u, v) for u, v in zip(got, want)) and ra.randn() == rb.randn()
    # long epochs under `sgd`: the permutations go into three arrays in turn, and a batch of an index-like data item (the
    # GLM's resident fit) is a VIEW of its permutation -- through a prefetch pipeline that holds its full complement of batches
    # (eight behind the first stage, one in and behind the second) every batch still arrives intact, epoch after epoch
    from revrand_amd.glm import _RowIndex
    N, M, steps = 70000, 4096, 75   # 17 batches per epoch, 4.4 epochs
    ra = np.random.RandomState(8)
    ref_perm, pos, want = np.empty(0, dtype=int), 0, []
    for _ in range(steps):
        parts, need = [], M
        while need:
            if pos == len(ref_perm):
                ref_perm, pos = ra.permutation(N), 0
            take = min(need, len(ref_perm) - pos)
            parts.append(ref_perm[pos:pos + take]); pos += take; need -= take
        want.append(np.concatenate(parts))

    class Loop(object):
        def begin(self, *a):
            self.seen = []

        def step(self, batch):
            import time
            time.sleep(0.002)   # a slow consumer: the queues fill up
            self.seen.append(np.array(batch[0], copy=True))

        def end(self):
            return np.zeros(1), np.zeros(len(self.seen)), np.zeros(len(self.seen))

        def abort(self):
            pass
    loop = Loop()
    opt.sgd(None, np.zeros(1), [_RowIndex(N)], batch_size=M, maxiter=steps, random_state=np.random.RandomState(8),
            prefetch=[lambda bt: list(bt), lambda bt: list(bt)], device_loop=loop)
    assert len(loop.seen) == steps and all(np.array_equal(u, v) for u, v in zip(loop.seen, want))


def test_schedule_model_of_the_pipelined_diagonal_block_cholesky():
    """tools/chol_diag_emu.py restates rr_chol_diag_pipe_kernel's schedule in NumPy (slots, early update of the slot that
    holds the next pivot row, deferred rest of the rank-1 update, compile-time column ranges, forward substitution in the
    lower slots): factor and inverse factor of a 128 x 128 block agree with numpy.linalg, and the next pivot is known from
    a running diagonal before its row arrives (the variant measured in DESIGN 3.10)."""
    import runpy
    ns = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "chol_diag_emu.py"))
    assert ns["ERR_U"] < 1e-13 and ns["ERR_UINV"] < 1e-13


def test_bench_line_is_numbers_only_and_small():
    """bench.py's ONE JSON line: keys starting with "_" (notes, samples, per-kernel detail) go to the full record only,
    floats carry 5 significant digits, a parity figure above its tolerance is an error -- never a measurement."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("rr_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = {"value": 8461234.56789, "n": np.int64(3), "x": np.float64(1.0) / 3, "nan": float("nan"), "flag": True,
           "_note": "prose " * 100, "roofline": {"frac": 0.94312345, "_what": "long", "list": [1.23456789e-7, 2]},
           "configs": {"a": {"_sample": "s", "ms": 12.3456789}}}
    line = json.dumps(bench.lean(rec))
    back = json.loads(line)
    assert back == {"value": 8461200.0, "n": 3, "x": 0.33333, "nan": None, "flag": True,
                    "roofline": {"frac": 0.94312, "list": [1.2346e-07, 2]}, "configs": {"a": {"ms": 12.346}}}
    fullrec = bench.full(rec)
    assert fullrec["note"].startswith("prose") and fullrec["configs"]["a"]["sample"] == "s" and fullrec["value"] == rec["value"]
    assert bench.parity("x", 1e-7, 1e-6) == 1e-7
    with pytest.raises(bench.ParityError, match="exceeds its tolerance"):
        bench.parity("Gram of 4096 rows vs oracle", 0.3, 1e-4)
    with pytest.raises(bench.ParityError):
        bench.parity("x", float("nan"), 1e-4)


def test_sgd_stops_its_prefetch_worker_before_an_exception_of_the_objective_leaves_it():
    """when `fun` raises inside `sgd`, the traceback keeps the prefetching generator -- and its worker thread -- alive
    past the caller's cleanup (glm.fit frees the device buffers the worker's next batch writes to).  `sgd` closes the
    generator itself: by the time the exception reaches the caller the worker has been told to stop and has been joined."""
    import threading
    import time
    from revrand_amd.optimize import sgd
    started, slow = [], threading.Event()

    def augment(batch):          # runs on the worker thread, one batch ahead of the step
        started.append(threading.current_thread())
        if len(started) >= 3:
            slow.set()
            time.sleep(0.3)      # the worker is in the middle of a batch when the objective raises
        return batch

    def fun(x, Xb):
        if slow.wait(5.0):
            raise RuntimeError("objective failed")
        return np.zeros_like(x)

    X = np.arange(200.0).reshape(100, 2)
    with pytest.raises(RuntimeError, match="objective failed"):
        sgd(fun, np.zeros(3), X, batch_size=10, maxiter=50, prefetch=augment, random_state=np.random.RandomState(0))
    assert started and not started[0].is_alive()
    # and a run that finishes normally leaves no worker either
    res = sgd(lambda x, Xb: np.ones_like(x), np.zeros(3), X, batch_size=10, maxiter=5, prefetch=lambda b: b,
              random_state=np.random.RandomState(0))
    assert res.x.shape == (3,) and threading.active_count() <= 2


# ---------------------------------------------------------------------------------------------------------------------
# "revrand_amd's bases work under the REFERENCE's unmodified estimator".  The reference is not
# on the GPU box and there is no GPU here, so the claim is tested at the protocol: the driver below makes exactly the
# calls revrand/slm.py makes on its basis -- :113 `basis.regularizer`, `basis.params`; :145 `basis.transform(X, *hypers)`;
# :150 `basis.regularizer_diagonal(X, *reg)`; :197 `apply_grad(dhyps, basis.grad(X, *hypers))`; :240 `basis.transform` of
# predict_moments -- and nothing else, on revrand_amd bases whose DEVICE handle is a stub that evaluates the oracle's
# formulas (what the kernels are tested against on the GPU).  Results: the reference's own `_elbo` outputs (golden).
# ---------------------------------------------------------------------------------------------------------------------

class _OracleRffHandle(object):
    """Stands in for _hip.RffHandle: same constructor and method signatures, the oracle's arithmetic."""
    calls = []

    def __init__(self, W, compute="f32", device=None):
        self.W = np.asarray(W, dtype=float)
        self.d, self.n = self.W.shape

    def transform(self, X, lenscale, out_dtype=np.float64):
        _OracleRffHandle.calls.append("transform")
        return orc.rff_transform(np.asarray(X, dtype=float), self.W, np.asarray(lenscale, dtype=float).squeeze())

    def grad(self, X, lenscale, out_dtype=np.float64):
        _OracleRffHandle.calls.append("grad")
        return orc.rff_grad(np.asarray(X, dtype=float), self.W, np.asarray(lenscale, dtype=float).squeeze())


def _reference_shaped_elbo(basis, X, y, var, reg, hypers):
    """revrand/slm.py:142-199, statement for statement in its use of `basis` (NumPy for everything else)."""
    from revrand_amd.basis_functions import apply_grad
    from revrand_amd.utils import atleast_list, issequence
    Phi = basis.transform(X, *atleast_list(hypers))                      # :145
    PhiPhi = Phi.T.dot(Phi)
    N, D = Phi.shape
    L, slices = basis.regularizer_diagonal(X, *atleast_list(reg))        # :150
    iL = 1. / L
    C = np.linalg.inv(np.diag(iL) + PhiPhi / var)
    m = C.dot(Phi.T.dot(y)) / var
    TrPhiPhiC = (PhiPhi * C).sum()
    Err = y - Phi.dot(m)
    sqErr = (Err ** 2).sum()
    logdetC = np.linalg.slogdet(C)[1]
    ELBO = -0.5 * (N * np.log(2 * np.pi * var) + sqErr / var + TrPhiPhiC / var + ((m ** 2 + C.diagonal()) * iL).sum()
                   - logdetC + np.log(L).sum() - D)
    dvar = 0.5 * (-N + (sqErr + TrPhiPhiC) / var) / var

    def dreg(s):
        return -0.5 * (((m[s] ** 2 + C[s, s].diagonal()) * iL[s] ** 2).sum() - iL[s].sum())
    dL = list(map(dreg, slices)) if issequence(slices) else dreg(slices)

    def dhyps(dPhi):
        return -(m.T.dot(Err.dot(dPhi)) - (dPhi.T.dot(Phi) * C).sum()) / var
    dhypers = apply_grad(dhyps, basis.grad(X, *atleast_list(hypers)))    # :197
    return ELBO, dvar, dL, dhypers, m, C


def test_bases_under_the_reference_estimators_call_sequence(golden, monkeypatch):
    import revrand_amd.basis_functions as bsm
    from revrand_amd import _hip
    monkeypatch.setattr(_hip, "RffHandle", _OracleRffHandle)
    g = golden("elbo")
    X, y, var = g["X"], g["y"], float(g["var"])
    d, n = X.shape[1], g["iso_W"].shape[1]
    cases = [("iso", bsm.RandomRBF(nbases=n, Xdim=d, random_state=21, lenscale=Parameter(1., Positive())), 1.7, float(g["iso_ls"])),
             ("ard", bsm.RandomRBF(nbases=n, Xdim=d, random_state=21, lenscale=Parameter(np.ones(d), Positive())), 1.7, g["ard_ls"]),
             ("cat", bsm.RandomMatern52(nbases=n, Xdim=d, random_state=22, lenscale=Parameter(np.ones(d), Positive()))
              + bsm.LinearBasis(onescol=True), [1.7, 0.6], g["cat_ls"])]
    for tag, basis, reg, hyp in cases:
        # :113 -- what the reference's fit hands to its optimiser
        params = [Parameter(1.0, Positive()), basis.regularizer, basis.params]
        assert len(params) == 3 and (tag != "cat" or len(basis.regularizer) == 2)
        _OracleRffHandle.calls.clear()
        ELBO, dvar, dL, dhyp, m, C = _reference_shaped_elbo(basis, X, y, var, reg, hyp)
        assert _OracleRffHandle.calls == ["transform", "grad"]           # one device call per protocol call, in its order
        assert abs(ELBO - float(g[tag + "_elbo"])) < 1e-10 * abs(float(g[tag + "_elbo"]))
        assert abs(dvar - float(g[tag + "_dvar"])) < 1e-8 * abs(float(g[tag + "_dvar"]))
        # (the fixtures hold the NEGATED entries of the list `_elbo` returns: [-dvar, dL, dhypers], slm.py:199)
        assert np.allclose(-np.atleast_1d(dL), g[tag + "_dreg"], rtol=1e-8)
        assert np.allclose(-np.atleast_1d(dhyp), g[tag + "_dhyp"], rtol=1e-7, atol=1e-9)
        assert np.allclose(m, g[tag + "_m"], rtol=1e-7, atol=1e-10) and np.allclose(C, g[tag + "_C"], rtol=1e-7, atol=1e-10)
        # :240-243 -- predict_moments' use of the basis
        Phi = basis.transform(X[:7], *([hyp] if tag != "cat" else [hyp]))
        assert Phi.shape == (7, m.size) and np.all(np.isfinite((Phi.dot(C) * Phi).sum(axis=1)))


# ---------------------------------------------------------------------------------------------------------------------
# revrand_amd/multigpu.py: the host logic of the in-process device group (no GPU: stub group / stub states)
# ---------------------------------------------------------------------------------------------------------------------

class _StubGroup(object):
    """DeviceGroup's interface without devices: `map` runs fn(i) in member order on this thread."""

    def __init__(self, n):
        import threading
        self.n, self._lock = n, threading.RLock()
        self.reduced = []

    def map(self, fn, members=None):
        return [fn(i) for i in (range(self.n) if members is None else members)]

    def reduce_stats(self, F, ptrs, nrows, wait=True):
        self.reduced.append((F, list(nrows)))
        return int(sum(nrows))


class _StubState(object):
    """A fit state over a row shard that computes with NumPy what the device state would (Phi given)."""

    def __init__(self, Phi, y):
        self.Phi, self.y, self.F, self.dev = Phi, y, Phi.shape[1], None
        self.acc = None
        self.dC = "C%d" % id(self)

    nrows = property(lambda self: self.Phi.shape[0])

    def gram_launch(self, hypers):
        self.launched = hypers

    def _stat_ptrs(self):
        return (None, None, None)

    def posterior(self, iL, var):
        return ("m", "dg", 1.0, 2.0)

    def second_pass(self, hypers, m, C, var):
        assert C == self.dC  # every member is handed ITS copy of the covariance
        r = self.y - self.Phi @ m
        return float(r @ r), [np.array([self.Phi.sum(), 1.0]), float(len(self.y))]

    def keep_best(self):
        self.kept = True

    def release(self):
        self.released = True


def test_multigpu_host_logic_shards_sums_and_routes():
    from revrand_amd import multigpu
    assert multigpu.resolve_devices([0, 0, 1]) == (0, 0, 1) and multigpu.resolve_devices(3) == (0, 1, 2)
    with pytest.raises(ValueError):
        multigpu.resolve_devices("some")
    with pytest.raises(ValueError):
        multigpu.resolve_devices(0)
    assert multigpu._tree_sum([[1.0, np.array([1.0, 2.0])], [2.0, np.array([3.0, 4.0])]])[1].tolist() == [4.0, 6.0]
    # ShardedFitState over stub states: contiguous shards equal to within a row, sums in member order, per-member covariance
    rs = np.random.RandomState(0)
    N, F, n = 103, 5, 4
    Phi, y, m = rs.randn(N, F), rs.randn(N), rs.randn(F)
    g = _StubGroup(n)
    bounds = [multigpu.shard_bounds(N, i, n) for i in range(n)]
    assert bounds[0][0] == 0 and bounds[-1][1] == N and max(e - s for s, e in bounds) - min(e - s for s, e in bounds) <= 1
    states = [_StubState(Phi[s:e], y[s:e]) for s, e in bounds]
    st = multigpu.ShardedFitState(g, states, bounds)
    assert st.N_total == N and st.F == F
    sq, dh = st.second_pass([1.0], m, st.dC, 0.5)
    assert abs(sq - float((y - Phi @ m) @ (y - Phi @ m))) < 1e-9
    assert abs(dh[0][0] - Phi.sum()) < 1e-9 and dh[0][1] == n and dh[1] == N
    assert st.posterior(np.ones(F), 0.5) == ("m", "dg", 1.0, 2.0)
    st.keep_best()
    assert st.best_on_device and all(s.kept for s in states)
    with pytest.raises(ValueError):
        st.gram_device([1.0], reduce=lambda *a: None)   # a process group on top of a device group: refused
    st.release()
    assert all(s.released for s in states)
    # minibatch routing of the sharded GLM features: every index goes to the member whose shard holds it, positions kept
    smf = multigpu.ShardedMinibatchFeatures.__new__(multigpu.ShardedMinibatchFeatures)
    smf.group, smf.n_use = g, 3
    smf.bounds = [multigpu.shard_bounds(1000, i, 3) for i in range(3)]
    smf.ends = np.array([e for _, e in smf.bounds])
    idx = rs.permutation(1000)[:200]
    parts = smf._split_idx(idx)
    seen = np.concatenate([pos for _, pos, _ in parts])
    assert sorted(seen.tolist()) == list(range(200))
    for i, pos, local in parts:
        s, e = smf.bounds[i]
        assert np.all((idx[pos] >= s) & (idx[pos] < e)) and np.array_equal(local, idx[pos] - s)
    assert [p[0] for p in smf._split_rows(5000)] == [0, 1] and [p[0] for p in smf._split_rows(100)] == [0]


def test_device_loops_of_a_group_host_logic():
    """glm._ScopedLoop (a one-device loop on one member of a device group: `devices=` with minibatches too small to split): every
    call runs with the member's context as the thread's default device, attribute reads and writes reach the inner loop -- what
    optimize.sgd / logtrick_sgd / structured_sgd do with a device loop (`log_coordinates = ...`, `getattr(loop, "note_start")`);
    glm._flat_params (a nested parameter list, FastFoodGM's [mean, lenscale] inside a concatenation's)."""
    from revrand_amd import _hip
    from revrand_amd import glm as G

    class Inner(object):
        log_coordinates = None

        def __init__(self):
            self.seen = []

        def begin(self, *a):
            self.seen.append(("begin", getattr(_hip._tls, "dev", None), a))
            return "began"

        def step(self, batch):
            self.seen.append(("step", getattr(_hip._tls, "dev", None), batch))
    inner, member = Inner(), object()
    before = getattr(_hip._tls, "dev", None)
    loop = G._ScopedLoop(inner, member)
    loop.log_coordinates = [True, False]
    assert inner.log_coordinates == [True, False] and loop.log_coordinates == [True, False]
    assert loop.begin(1, 2) == "began" and loop.step([3]) is None
    assert [(n, d is member) for n, d, _ in inner.seen] == [("begin", True), ("step", True)]
    assert getattr(_hip._tls, "dev", None) is before                       # restored after every call
    assert getattr(loop, "note_start", None) is None                        # absent on the inner loop: absent here
    assert isinstance(getattr(loop, "_loop"), Inner)                        # (how `fit` tells a fused loop from a step-per-call one)
    a, b, c = Parameter(1.0, Positive()), Parameter(np.ones(3), Bound()), Parameter(np.ones(3), Positive())
    assert G._flat_params([a, [b, c], []]) == [a, b, c] and G._flat_params(a) == [a]


def test_parameter_draws_equal_scipys_without_its_overhead():
    """btypes.Parameter.rvs (btypes.py:290-324) for the frozen `norm` / `gamma` distributions the estimators use goes straight
    to RandomState.standard_normal / standard_gamma -- the calls scipy's rvs ends in: same values, same type, same stream
    afterwards; anything else is scipy's own rvs."""
    from scipy.stats import gamma, norm, uniform
    from revrand_amd.btypes import Bound, Parameter, Positive, _frozen_rvs
    for dist in (norm(), norm(1.5, 0.3), norm(loc=-2, scale=4), gamma(1.), gamma(a=2, scale=0.5), gamma(4., scale=0.25),
                 gamma(3., loc=1.0, scale=2.0)):
        for shape in ((), (3,), (7, 4), (83, 10)):
            r1, r2 = np.random.RandomState(5), np.random.RandomState(5)
            a, b = _frozen_rvs(dist, shape, r1), dist.rvs(size=shape, random_state=r2)
            assert a is not None and type(a) is type(b) and np.array_equal(np.asarray(a), np.asarray(b))
            assert r1.randn() == r2.randn()
            for bounds in (Positive(), Bound(0.2, 1.1)):
                p = Parameter(dist, bounds, shape=shape)
                r1, r2 = np.random.RandomState(9), np.random.RandomState(9)
                assert np.array_equal(np.asarray(p.rvs(r1)), np.asarray(bounds.clip(dist.rvs(size=shape, random_state=r2))))
    assert _frozen_rvs(uniform(), (2,), np.random.RandomState(0)) is None          # not one of the two: scipy's rvs
    assert _frozen_rvs(norm(), (2,), np.random.default_rng(0)) is None             # not a legacy RandomState
    p = Parameter(uniform(0.5, 1.0), Positive(), shape=(3,))
    assert np.array_equal(p.rvs(np.random.RandomState(3)), uniform(0.5, 1.0).rvs(size=(3,), random_state=np.random.RandomState(3)))
