"""The in-process device group ACROSS physical GPUs -- and the parts of it a one-GPU box can still execute.

tests/test_gpu_multigpu.py runs every group on members that SHARE the box's one GPU (``devices=[0, 0]``): the peer
transport's kernels then load "peer" buffers from local HBM and RCCL is never entered (it refuses two ranks on one device).
Here:

* with >= 2 visible GPUs (skipped otherwise -- the first multi-GPU box switches them on by itself): device lists
  ``list(range(min(k, visible)))`` for k = 2, 4, 8 under BOTH transports -- the collectives against NumPy, the sharded
  statistics / `_elbo` / `fit` / `predict_moments` / GLM against the one-context results, members bit-identical, and
  peer == RCCL == one context;
* on any box: a ONE-member group under ``transport="rccl"`` goes through ``ncclCommInitAll`` and
  ``ncclGroupStart -> ncclAllReduce -> ncclGroupEnd`` / ``ncclBroadcast`` (rr_comm.hip: group_rccl_allreduce) -- the symbols
  and the call order of the N-member case; ``RR_TRANSPORT_AUTO``'s probe collective and its fall-back to the peer
  transport (forced by RR_COMM_PROBE_FAIL); the debug build's check that every launch is made with its stream's device
  current (tests/test_debug_builds.py).
Reference: one `fit` call in one process, revrand/slm.py:74-140; what the members sum is slm.py:145-157's statistics."""
import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise

pytestmark = pytest.mark.gpu


def _visible():
    try:
        from revrand_amd import multigpu
        return multigpu.visible_devices()
    except Exception:
        return 0


VISIBLE = _visible()
LISTS = sorted({tuple(range(min(k, VISIBLE))) for k in (2, 4, 8)} - {(0,), ()}, key=len) if VISIBLE >= 2 else []
CASES = [(list(d), t) for d in LISTS for t in ("peer", "rccl")]
IDS = ["%dgpu-%s" % (len(d), t) for d, t in CASES]
need2 = pytest.mark.skipif(VISIBLE < 2, reason="needs >= 2 visible GPUs (this box has %d)" % VISIBLE)


def _setup():
    import revrand_amd.basis_functions as bs
    from revrand_amd import _hip, multigpu
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    return bs, _hip, multigpu, Parameter, Positive, StandardLinearModel


def _data(N, d, seed=0):
    rs = np.random.RandomState(seed)
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)).astype(np.float32)
    return X, y


# ---- any box ---------------------------------------------------------------------------------------------------------------

def test_one_member_group_under_rccl_runs_the_grouped_calls(monkeypatch):
    """devices=[0], transport "rccl": ncclCommInitAll over one device, then every collective as GroupStart / per-member call /
    GroupEnd.  Sum, max, min over one member and a broadcast from it leave the buffer as it is; the statistics exchange of
    `gram` (pack -> all-reduce -> unpack + mirror) gives the one-context statistics bit for bit."""
    bs, _hip, multigpu, *_ = _setup()
    g = multigpu.DeviceGroup([0], transport="rccl")
    try:
        assert g.transport == "rccl" and g.n == 1
        v = np.random.RandomState(0).randn(5001)
        for op in ("sum", "max", "min"):
            buf = g.members[0].upload_vector(v)
            g.allreduce_device([buf], v.size, op)
            assert np.array_equal(g.members[0].download(buf, v.shape, np.float64), v)
            buf.free()
        buf = g.members[0].upload_vector(v)
        g.broadcast_device([buf], v.size * 8, 0)
        assert np.array_equal(g.members[0].download(buf, v.shape, np.float64), v)
        buf.free()
    finally:
        g.close()
    monkeypatch.setenv("RR_COMM_TRANSPORT", "rccl")
    X, y = _data(5000, 6)
    basis = bs.RandomRBF(nbases=96, Xdim=6, random_state=1)
    dev = _hip.get_device()
    was = dev.set_deterministic(True)
    try:
        G1, b1, t1 = basis.gram(X, y, 1.1)
        Gg, bg, tg = basis.gram(X, y, 1.1, devices=[0])
        assert multigpu.get_group([0]).transport == "rccl"
    finally:
        dev.set_deterministic(was)
    assert normwise(Gg, G1) < 1e-12 and normwise(bg, b1) < 1e-12 and abs(tg - t1) <= 1e-12 * abs(t1)
    assert np.array_equal(Gg, Gg.T)


def test_auto_transport_probes_rccl_and_falls_back_to_peer(monkeypatch, capfd):
    """RR_TRANSPORT_AUTO with RCCL preferred: the group's first grouped all-reduce is a probe at creation; when it fails
    (forced here) the group says so and takes the peer transport -- and works."""
    bs, _hip, multigpu, *_ = _setup()
    monkeypatch.setenv("RR_COMM_TRANSPORT", "rccl")
    g = multigpu.DeviceGroup([0], transport="auto")    # the probe passes: RCCL
    try:
        assert g.transport == "rccl"
    finally:
        g.close()
    monkeypatch.setenv("RR_COMM_PROBE_FAIL", "1")
    g = multigpu.DeviceGroup([0], transport="auto")
    try:
        assert g.transport == "peer"
        assert "uses the peer transport" in capfd.readouterr().err
        v = np.arange(100.0)
        buf = g.members[0].upload_vector(v)
        g.allreduce_device([buf], 100)
        assert np.array_equal(g.members[0].download(buf, v.shape, np.float64), v)
        buf.free()
    finally:
        g.close()
    with pytest.raises(_hip.HipError):                  # an EXPLICIT rccl request keeps its error
        monkeypatch.delenv("RR_COMM_TRANSPORT")
        multigpu.DeviceGroup([0, 0], transport="rccl")


# ---- >= 2 GPUs -------------------------------------------------------------------------------------------------------------

@need2
@pytest.mark.parametrize("devices,transport", CASES, ids=IDS)
def test_collectives_between_gpus_vs_numpy(devices, transport):
    """rr_comm_group_allreduce_dev / _broadcast_dev with every member on its own GPU.  Peer transport: bit-equal to NumPy's
    sum in member order.  RCCL: its own order -- equal to 1e-15 of the largest term, and all members bit-identical."""
    bs, _hip, multigpu, *_ = _setup()
    n = len(devices)
    g = multigpu.DeviceGroup(devices, transport=transport)
    try:
        assert g.transport == transport and g.n == n
        rs = np.random.RandomState(n)
        for count in (1, 31, 33, 4099, (1 << 20) + 3, 34_100_000 // 8):
            host = [rs.randn(count) * 10.0 ** rs.randint(-3, 4) for _ in range(n)]
            for op, ref in (("sum", None), ("max", np.maximum), ("min", np.minimum)):
                bufs = [m.upload_vector(h) for m, h in zip(g.members, host)]
                g.allreduce_device(bufs, count, op)
                want = host[0].copy()
                for h in host[1:]:
                    want = (want + h) if ref is None else ref(want, h)
                got = [m.download(b, (count,), np.float64) for m, b in zip(g.members, bufs)]
                for b in bufs:
                    b.free()
                for o in got[1:]:
                    assert np.array_equal(o, got[0]), (n, count, op, "members differ")
                if ref is not None or transport == "peer":
                    assert np.array_equal(got[0], want), (n, count, op)
                else:
                    scale = np.max([np.abs(h) for h in host], axis=0)
                    assert np.all(np.abs(got[0] - want) <= 4e-16 * n * scale), (n, count, op)
        count = 70001
        host = [rs.randn(count) for _ in range(n)]
        for root in sorted({0, n - 1}):
            bufs = [m.upload_vector(h) for m, h in zip(g.members, host)]
            g.broadcast_device(bufs, count * 8, root)
            for m, b in zip(g.members, bufs):
                assert np.array_equal(m.download(b, (count,), np.float64), host[root])
                b.free()
    finally:
        g.close()


@need2
@pytest.mark.parametrize("devices,transport", CASES, ids=IDS)
def test_sharded_statistics_between_gpus(devices, transport, monkeypatch):
    """basis.gram(..., devices=) with the rows on several GPUs: the full matrices against the one-context call and the
    oracle, single basis and config 3's concatenation, ragged row counts."""
    bs, _hip, multigpu, *_ = _setup()
    monkeypatch.setenv("RR_COMM_TRANSPORT", transport)
    X, y = _data(50021, 8)
    basis = bs.RandomRBF(nbases=160, Xdim=8, random_state=1)
    ls = np.linspace(0.7, 1.6, 8)
    G1, b1, t1 = basis.gram(X, y, ls)
    G, b, t = basis.gram(X, y, ls, devices=devices)
    assert multigpu.get_group(devices).transport == transport
    assert normwise(G, G1) < 1e-6 and normwise(b, b1) < 1e-6 and abs(t - t1) < 1e-6 * t1
    Gr, br, tr = orc.gram_stats(orc.rff_transform(X, basis.W, ls), y.astype(np.float64))
    assert normwise(G, Gr) < 5e-6 and normwise(b, br) < 5e-5 and np.array_equal(G, G.T)
    sl = slice(0, 2 * len(devices) + 1)   # fewer rows than make a tile: some members get two or three rows
    Gs, bsub, _ = basis.gram(X[sl], y[sl], ls, devices=devices)
    assert normwise(Gs, orc.gram_stats(orc.rff_transform(X[sl], basis.W, ls), y[sl].astype(np.float64))[0]) < 5e-6
    cat = bs.RandomMatern52(nbases=96, Xdim=8, random_state=2) + bs.LinearBasis(onescol=True)
    Gc1, bc1, _ = cat.gram(X, y, 1.3)
    Gc, bc, _ = cat.gram(X, y, 1.3, devices=devices)
    assert normwise(Gc, Gc1) < 1e-6 and normwise(bc, bc1) < 1e-6


def _elbo_once(SLM, basis, X, y, var, reg, ls, devices=None):
    """One `_elbo` on a fresh fit state; with devices: every member's copy of the statistics and of the covariance too."""
    from revrand_amd.utils import flatten_values
    slm = SLM(basis, devices=devices)
    slm.obj_ = -np.inf
    slm._state = slm._make_state(X, y)
    assert slm._state is not None
    f, (gv, gr, gh) = slm._elbo(X, y, var, reg, ls)
    extra = None
    if devices is not None:
        st = slm._state
        extra = ([s.stats_host() for s in st.states], [s.dev.download(s.dCbest, (s.F, s.F), np.float64) for s in st.states])
    slm._state.release()
    slm._state = None
    return np.asarray(flatten_values([f, gv, gr, gh]), dtype=float), np.array(slm.weights_), extra


@need2
@pytest.mark.parametrize("devices", [list(d) for d in LISTS], ids=["%dgpu" % len(d) for d in LISTS])
def test_peer_rccl_and_one_context_agree_and_members_are_identical(devices, monkeypatch):
    """Deterministic mode, posterior on the device: every member's partial statistics are reproducible, so the two
    transports differ only in the order of ONE float64 sum over the members -- and each leaves all members with
    bit-identical statistics and posterior covariance."""
    bs, _hip, multigpu, Parameter, Positive, SLM = _setup()
    monkeypatch.setenv("RR_POSDEF", "device")
    X, y = _data(40000, 6, seed=3)
    basis = bs.RandomRBF(nbases=128, Xdim=6, random_state=1, lenscale=Parameter(np.ones(6), Positive()))
    ls = np.linspace(0.8, 1.4, 6)
    out = {}
    for transport in ("peer", "rccl"):
        monkeypatch.setenv("RR_COMM_TRANSPORT", transport)
        g = multigpu.get_group(devices)
        assert g.transport == transport
        was = g.set_deterministic(True)
        try:
            out[transport] = _elbo_once(SLM, basis, X, y, 0.3, 1.2, ls, devices=devices)
        finally:
            for m, w in zip(g.members, was):
                m.set_deterministic(w)
        stats, covs = out[transport][2]
        for (G, b, t), C in zip(stats[1:], covs[1:]):
            assert np.array_equal(G, stats[0][0]) and np.array_equal(b, stats[0][1]) and t == stats[0][2], transport
            assert np.array_equal(C, covs[0]), transport
    v1, w1, _ = _elbo_once(SLM, basis, X, y, 0.3, 1.2, ls)
    for t in ("peer", "rccl"):
        v, w, _ = out[t]
        assert abs(v[0] - v1[0]) < 1e-6 * abs(v1[0]) and normwise(v[1:], v1[1:]) < 1e-4 and normwise(w, w1) < 2e-4, t
    assert normwise(out["peer"][2][0][0][0], out["rccl"][2][0][0][0]) < 1e-14      # G: one sum over the members, reordered
    assert abs(out["peer"][0][0] - out["rccl"][0][0]) < 1e-9 * abs(v1[0])
    assert normwise(out["peer"][0][1:], out["rccl"][0][1:]) < 1e-7


@need2
@pytest.mark.parametrize("devices,transport", CASES, ids=IDS)
def test_fit_predict_and_glm_between_gpus(devices, transport, monkeypatch):
    """`fit` / `predict_moments` of the SLM and a GLM fit with devices= on distinct GPUs against the one-context runs."""
    bs, _hip, multigpu, Parameter, Positive, SLM = _setup()
    from revrand_amd.glm import GeneralizedLinearModel
    from revrand_amd import likelihoods as lk
    monkeypatch.setenv("RR_COMM_TRANSPORT", transport)
    X, y = _data(30000, 5, seed=1)

    def make(dv):
        basis = bs.RandomMatern52(nbases=96, Xdim=5, random_state=2, lenscale=Parameter(np.ones(5), Positive())) \
            + bs.LinearBasis(onescol=True)
        return SLM(basis, nstarts=0, maxiter=20, random_state=0, devices=dv)
    one, many = make(None).fit(X, y), make(devices).fit(X, y)
    assert multigpu.get_group(devices).transport == transport
    assert abs(many.obj_ - one.obj_) < 1e-4 * abs(one.obj_)
    assert normwise(many.weights_, one.weights_) < 5e-3
    Ey1, Vy1 = one.predict_moments(X[:5000])
    Ey, Vy = many.predict_moments(X[:5000])
    assert normwise(Ey, Ey1) < 5e-3 and normwise(Vy, Vy1) < 5e-3
    # the same fitted model served by the group: rows are independent, the result is the one-context one
    many.weights_, many.covariance_, many.hypers_, many.var_ = one.weights_, one.covariance_, one.hypers_, one.var_
    many._drop_serving()
    Ey2, Vy2 = many.predict_moments(X[:5000])
    assert np.array_equal(Ey2, Ey1) and np.array_equal(Vy2, Vy1)

    rs = np.random.RandomState(5)
    yp = rs.poisson(np.exp(0.4 * np.sin(X[:, 0]))).astype(float)

    # the GLM: the resident loop over the group (rr_glm_sgd_group_step: every member's share of a minibatch on its own GPU, the
    # row sums all-reduced across them, the update replicated) and the host loop around the sharded `_elbo`
    from revrand_amd import _hip as hip_
    steps = [0]
    real = hip_.ResidentSgdGroup.step

    def spy(self, *a, **k):
        steps[0] += 1
        return real(self, *a, **k)
    monkeypatch.setattr(hip_.ResidentSgdGroup, "step", spy)

    def glm(dv, resident=True):
        basis = bs.RandomRBF(nbases=64, Xdim=5, random_state=3, lenscale=Parameter(np.ones(5), Positive()))
        np.random.seed(4)
        g = GeneralizedLinearModel(lk.Poisson(), basis, K=3, nsamples=8, batch_size=2048 * len(devices), maxiter=6, nstarts=2,
                                   random_state=2, devices=dv)
        g._resident_sgd = resident
        return g.fit(X, yp)
    a, b, c = glm(None), glm(devices), glm(devices, resident=False)
    assert steps[0] == 6
    for m in (b, c):
        assert normwise(m.weights_, a.weights_) < 1e-4 and normwise(m.covariance_, a.covariance_) < 1e-4
        assert normwise(m.basis_hypers_, a.basis_hypers_) < 1e-4
    assert a.random_.randn() == b.random_.randn() == c.random_.randn()


# ---- every kind of resident fit state behind devices= (distinct GPUs when the box has two, else two members on one) -------

DEV2 = [0, 1] if VISIBLE >= 2 else [0, 0]


@pytest.mark.parametrize("kind", ["fastfood", "fastfood_gm", "rbf_f64", "cat_f64", "config4_width"])
def test_elbo_with_devices_for_every_fit_state(kind, monkeypatch):
    """`_elbo` (objective, all gradients, weights) with the rows on two members against one context for the fit states the
    round-5 tests did not cover: FastFoodRBF and FastFoodGM (the chain kernels writing into each member's feature matrix),
    dtype="f64" (the float64 pipeline end to end: 1e-9) alone and in a concatenation, and config 4's `_elbo` width
    (FastFoodRBF F = 16384, D = 128: SURVEY 8e lists C4 as row-sharded too)."""
    bs, _hip, multigpu, Parameter, Positive, SLM = _setup()
    from revrand_amd.btypes import Bound
    from revrand_amd.utils import flatten_values
    monkeypatch.setenv("RR_POSDEF", "device")
    d = 128 if kind == "config4_width" else (16 if kind.startswith("fastfood") else 6)   # (the GM chain wants d2 >= 16)
    N = 6000 if kind == "config4_width" else 30001
    X, y = _data(N, d, seed=9)
    ard = lambda: Parameter(np.ones(d), Positive())  # noqa: E731
    ls = np.linspace(0.8, 1.3, d)
    if kind == "fastfood":
        basis, reg, hyp, tol = bs.FastFoodRBF(nbases=96, Xdim=d, random_state=2, lenscale=ard()), 1.2, ls, 2e-4
    elif kind == "config4_width":
        basis, reg, hyp, tol = bs.FastFoodRBF(nbases=8192, Xdim=d, random_state=2, lenscale=ard()), 1.2, ls * 6.0, 2e-4
    elif kind == "fastfood_gm":
        basis = bs.FastFoodGM(nbases=96, Xdim=d, random_state=2, mean=Parameter(np.zeros(d), Bound()), lenscale=ard())
        reg, hyp, tol = 1.2, [0.1 * np.sin(np.arange(d)), ls], 2e-4
    elif kind == "rbf_f64":
        basis, reg, hyp, tol = bs.RandomRBF(nbases=80, Xdim=d, random_state=2, lenscale=ard(), dtype="f64"), 1.2, ls, 1e-9
    else:
        basis = bs.RandomRBF(nbases=80, Xdim=d, random_state=2, lenscale=ard(), dtype="f64") + bs.LinearBasis(onescol=True)
        reg, hyp, tol = [1.2, 0.7], ls, 1e-9
    Xf = X.astype(np.float64) if "f64" in kind else X

    def once(devices):
        slm = SLM(basis, devices=devices)
        slm.obj_ = -np.inf
        slm._state = slm._make_state(Xf, y.astype(Xf.dtype))
        assert slm._state is not None          # resident: the statistics never leave HBM
        try:
            f, g = slm._elbo(Xf, y, 0.3, reg, hyp)
        finally:
            slm._state.release()
            slm._state = None
        return np.asarray(flatten_values([f] + list(g)), dtype=float), np.array(slm.weights_)
    v1, w1 = once(None)
    v2, w2 = once(DEV2)
    assert v1.shape == v2.shape and np.all(np.isfinite(v2))
    assert abs(v2[0] - v1[0]) < max(tol * 1e-2, 1e-12) * abs(v1[0]), (v2[0], v1[0])
    assert normwise(v2[1:], v1[1:]) < tol and normwise(w2, w1) < 2 * tol
