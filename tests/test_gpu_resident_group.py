"""`GeneralizedLinearModel(devices=[...]).fit` with its SVI loop resident on EVERY member of the device group
(glm._GroupResidentLoop over rr_glm_sgd_group_step): the minibatch's rows are split by owner, every member forms the step's
products on its rows, the three row sums (length-scale contractions, [Edm | EdC], likelihood sums) are all-reduced in HBM and
every member makes the same update of its copy of the parameters -- against the ONE-context resident loop, itself held to the
reference's fits by tests/test_gpu_glm_fit.py (reference: one `fit` call, revrand/glm.py:141-203; what a step sums over
rows is glm.py:229-283).  Members share the box's one GPU here (``devices=[0, 0]``); tests/test_gpu_multigpu_devices.py runs
the same loop across distinct GPUs when the box has them.  Same seeds -> same minibatches, draws and start point."""
import numpy as np
import pytest

from conftest import normwise

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["two streams", "one stream"])
def _order_of_work(monkeypatch, request):
    monkeypatch.setenv("RR_GLM_SGD_OVERLAP", "1" if request.param == "two streams" else "0")


def _imports():
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    return bs, lk, Parameter, Positive, GeneralizedLinearModel


def _data(lik, N=6000, d=5, seed=4):
    rs = np.random.RandomState(seed)
    X = rs.randn(N, d)
    f = 0.5 * np.sin(X[:, 0]) + 0.2 * X[:, 2]
    if lik == "poisson":
        return X, rs.poisson(np.exp(f)).astype(float), ()
    if lik == "bernoulli":
        return X, (rs.rand(N) < 1 / (1 + np.exp(-3 * f))).astype(float), ()
    if lik == "binomial":
        n = rs.randint(5, 30, size=N).astype(float)
        return X, rs.binomial(n.astype(int), 1 / (1 + np.exp(-3 * f))).astype(float), (n,)
    return X, f + 0.1 * rs.randn(N), ()


def _basis(kind, d):
    bs, lk, Parameter, Positive, GLM = _imports()
    if kind == "ard":
        return bs.RandomRBF(nbases=48, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    if kind == "iso":
        return bs.RandomMatern32(nbases=48, Xdim=d, random_state=1)
    if kind == "gm":   # a spectral-mixture component (two parameters per input dimension) next to a linear child
        return bs.LinearBasis(onescol=True) + bs.FastFoodGM(nbases=16, Xdim=d, random_state=2)
    return bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=24, Xdim=d, random_state=2) \
        + bs.FastFoodRBF(nbases=16, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive()))


def _fit(devices, lik="poisson", basis="ard", sampler="host", batch=1500, maxiter=12, nstarts=2, N=6000, fused=True, K=3):
    bs, lk, Parameter, Positive, GLM = _imports()
    X, y, largs = _data(lik, N=N, d=12 if basis == "gm" else 5)
    like = {"poisson": lk.Poisson, "bernoulli": lk.Bernoulli, "binomial": lk.Binomial, "gaussian": lk.Gaussian}[lik]()
    glm = GLM(like, _basis(basis, X.shape[1]), K=K, nsamples=8, batch_size=batch, maxiter=maxiter, nstarts=nstarts, random_state=11,
              sampler=sampler, devices=devices)
    glm._fused_sgd = fused
    np.random.seed(3)  # (the start point is a draw from NumPy's global stream, as in the reference)
    glm.fit(X, y, likelihood_args=largs)
    return (glm.weights_.copy(), glm.covariance_.copy(), _flat(glm.regularizer_), _flat(glm.like_hypers_), _flat(glm.basis_hypers_),
            glm.random_.randn())


def _flat(v):
    if isinstance(v, (list, tuple)):
        return np.concatenate([_flat(u) for u in v]) if len(v) else np.empty(0)
    return np.atleast_1d(np.asarray(v, dtype=float)).ravel()


def _same(a, b, tol):
    for u, v in zip(a[:5], b[:5]):
        assert u.shape == v.shape
        if u.size:
            assert normwise(u, v) < tol
    assert a[5] == b[5]  # the RandomState ends where the one-context fit's does


@pytest.fixture
def spies(monkeypatch):
    """Counts the steps each kind of loop takes, keeps every member's parameters as the group loop leaves them and the rows each
    member got per step; members take part from 256 rows of a minibatch each (the library's 2048 would need larger data)."""
    from revrand_amd import _hip, multigpu
    monkeypatch.setattr(multigpu.ShardedMinibatchFeatures, "MIN_ROWS_PER_MEMBER", 256)
    seen = {"group": 0, "one": 0, "fused": 0, "z": [], "rows": []}
    real_g, real_1, real_f, real_close = _hip.ResidentSgdGroup.step, _hip.ResidentSgd.step, _hip.FusedSvi.run, _hip.ResidentSgdGroup.close

    def g(self, parts, *a, **k):
        seen["group"] += 1
        seen["rows"].append([p[1] for p in parts])
        return real_g(self, parts, *a, **k)

    def one(self, *a, **k):
        seen["one"] += 1
        return real_1(self, *a, **k)

    def f(self, n, *a, **k):
        seen["fused"] += n
        return real_f(self, n, *a, **k)

    def close(self):
        if all(s.h for s in self.sgds):
            seen["z"].append([s.read()[0] for s in self.sgds])
        return real_close(self)
    monkeypatch.setattr(_hip.ResidentSgdGroup, "step", g)
    monkeypatch.setattr(_hip.ResidentSgd, "step", one)
    monkeypatch.setattr(_hip.FusedSvi, "run", f)
    monkeypatch.setattr(_hip.ResidentSgdGroup, "close", close)
    return seen


@pytest.mark.parametrize("sampler", ["host", "device"])
@pytest.mark.parametrize("lik,basis,devices", [("poisson", "ard", [0, 0]), ("binomial", "iso", [0, 0, 0]), ("gaussian", "cat", [0, 0]),
                                               ("bernoulli", "cat", [0, 0, 0, 0]), ("poisson", "gm", [0, 0])])
def test_group_resident_fit_equals_the_one_context_fit(lik, basis, devices, sampler, spies):
    one = _fit(None, lik, basis, sampler)
    assert spies["one"] == 12 and spies["group"] == 0
    many = _fit(devices, lik, basis, sampler)
    assert spies["group"] == 12 and spies["one"] == 12      # every step of the second fit went through the group's loop
    assert all(sum(r) == 1500 and len(r) == len(devices) for r in spies["rows"])
    _same(many, one, 2e-5)   # (the step's float32 K-split atomics and the split of the row sums: tests/test_gpu_resident_sgd.py)
    zs = spies["z"][-1]
    assert len(zs) == len(devices) and all(np.array_equal(z, zs[0]) for z in zs[1:])   # the members' copies: the same bits


@pytest.mark.parametrize("switch", ["RR_GLM_BATCH_PREFETCH", "RR_GLM_DRAW_UPLOAD", "RR_GLM_PREFETCH_STAGES"])
def test_group_loop_without_the_worker_s_uploads(switch, spies, monkeypatch):
    """The step uploads its indices / targets / draws itself when the minibatch worker did not (measurement switches): same fit."""
    one = _fit(None, "binomial", "ard", "host")
    monkeypatch.setenv(switch, "1" if switch.endswith("STAGES") else "0")
    many = _fit([0, 0], "binomial", "ard", "host")
    assert spies["group"] == 12
    _same(many, one, 2e-5)


def test_a_member_without_rows_of_a_minibatch_follows_the_others(spies, monkeypatch):
    """Three members, three rows per minibatch: most steps leave a member without rows (rr_glm_sgd_group_step: rows == 0) -- its
    sums are zero, its parameters those of the others."""
    from revrand_amd import multigpu
    monkeypatch.setattr(multigpu.ShardedMinibatchFeatures, "MIN_ROWS_PER_MEMBER", 1)
    one = _fit(None, "poisson", "ard", "device", batch=3, maxiter=25, nstarts=0, N=30, fused=False)
    many = _fit([0, 0, 0], "poisson", "ard", "device", batch=3, maxiter=25, nstarts=0, N=30, fused=False)
    assert spies["group"] == 25 and any(0 in r for r in spies["rows"])
    _same(many, one, 2e-5)
    zs = spies["z"][-1]
    assert all(np.array_equal(z, zs[0]) for z in zs[1:])


def test_minibatches_too_small_to_split_run_the_one_device_loops_on_member_0(spies):
    """The reference's default batch_size = 10 under devices=: the fused small-batch loop (rr_glm_svi) on member 0's context --
    the same kernels on the same GPU as without devices=, so the same fit to rounding -- and a mid-sized minibatch (below two
    members' minimum) through rr_glm_sgd there."""
    one = _fit(None, "poisson", "cat", "host", batch=10, maxiter=40, nstarts=3)
    n_fused = spies["fused"]
    assert n_fused == 40
    many = _fit([0, 0], "poisson", "cat", "host", batch=10, maxiter=40, nstarts=3)
    assert spies["fused"] == 2 * n_fused and spies["group"] == 0
    _same(many, one, 1e-9)
    one = _fit(None, "gaussian", "ard", "device", batch=300, maxiter=10)
    many = _fit([0, 0], "gaussian", "ard", "device", batch=300, maxiter=10)
    assert spies["one"] == 20 and spies["group"] == 0
    _same(many, one, 1e-6)
