"""rr_glm_svi / glm._FusedLoop: the SVI loop of small minibatches as ONE persistent kernel (many SGD steps per launch, K
cooperating workgroups) and structured_sgd's random starts as one launch -- against the loops it stands in for.  The
reference's fits themselves (tests/golden/glm_fit.npz) are held against it by tests/test_gpu_glm_fit.py; here:

* every likelihood, every updater, concatenations, isotropic / ARD / bounded length scales, both samplers: fused == the
  step-per-call resident loop (rr_glm_sgd_step) == the host loop around `_elbo` on the same minibatches and draws
  (reference: optimize/sgd.py:337-425, decorators.py:133-252, 329-408, 541-583, glm.py:205-322);
* the batched random starts pick the candidate the sequential evaluation picks, from the same objective values;
* a fit cut into several launches (blocks of steps) is bit-identical to the same fit in one launch -- the kernel's sums
  have a fixed order -- and two runs are bit-identical;
* the shapes rr_glm_svi_supported declines go to the step-per-call loop."""
import logging

import numpy as np
import pytest

from conftest import normwise

pytestmark = pytest.mark.gpu


def _imports():
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd import optimize as opt
    from revrand_amd.btypes import Bound, Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    return bs, lk, opt, Bound, Parameter, Positive, GeneralizedLinearModel


def _data(lik, N=500, d=4, seed=4):
    rs = np.random.RandomState(seed)
    X = rs.randn(N, d)
    f = 0.5 * np.sin(X[:, 0]) + 0.2 * X[:, 2]
    if lik in ("poisson", "poisson_softplus"):
        return X, rs.poisson(np.exp(f)).astype(float), ()
    if lik == "bernoulli":
        return X, (rs.rand(N) < 1 / (1 + np.exp(-3 * f))).astype(float), ()
    if lik == "binomial":
        n = rs.randint(5, 30, size=N).astype(float)
        return X, rs.binomial(n.astype(int), 1 / (1 + np.exp(-3 * f))).astype(float), (n,)
    return X, f + 0.1 * rs.randn(N), ()


def _fit(loop, lik="poisson", updater=None, basis="ard", sampler="host", maxiter=25, K=4, L=10, batch=10, nstarts=3, block=None,
         record=None):
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    from revrand_amd import _hip
    from revrand_amd import glm as glm_mod
    X, y, largs = _data(lik)
    d = X.shape[1]
    if basis == "ard":
        b = bs.RandomRBF(nbases=12, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    elif basis == "iso":
        b = bs.RandomRBF(nbases=12, Xdim=d, random_state=1)
    elif basis == "bound":
        b = bs.RandomRBF(nbases=12, Xdim=d, random_state=1, lenscale=Parameter(1.0, Bound(0.995, 1.004)))
    elif basis == "cat":
        b = bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=10, Xdim=d, random_state=1) \
            + bs.RandomMatern52(nbases=6, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive()))
    elif basis == "cat_ind":   # children on their own columns of X (apply_ind), a linear child without the column of ones
        b = bs.RandomRBF(nbases=8, Xdim=2, random_state=1, apply_ind=[0, 2]) + bs.LinearBasis(onescol=False, apply_ind=[1, 3])
    like = {"poisson": lambda: lk.Poisson(), "poisson_softplus": lambda: lk.Poisson("softplus"), "bernoulli": lk.Bernoulli,
            "binomial": lk.Binomial, "gaussian": lk.Gaussian}[lik]()
    glm = GLM(like, b, K=K, nsamples=L, batch_size=batch, maxiter=maxiter, nstarts=nstarts, random_state=11,
              updater=updater() if updater is not None else None, sampler=sampler)
    glm._resident_sgd = loop != "host"
    glm._fused_sgd = loop == "fused"
    if block is not None:
        old = glm_mod._FusedLoop.BLOCK_STEPS
        glm_mod._FusedLoop.BLOCK_STEPS = block
    calls = {"run": 0, "steps": 0, "starts": 0, "resident": 0}
    real_run, real_starts, real_step = _hip.FusedSvi.run, _hip.FusedSvi.starts, _hip.ResidentSgd.step

    def run(self, n, *a, **k):
        calls["run"] += 1
        calls["steps"] += n
        return real_run(self, n, *a, **k)

    def starts(self, didx, cand, *a, **k):
        calls["starts"] += len(cand)
        out = real_starts(self, didx, cand, *a, **k)
        if record is not None:
            record.append(np.array(out))
        return out

    def step(self, *a, **k):
        calls["resident"] += 1
        return real_step(self, *a, **k)
    _hip.FusedSvi.run, _hip.FusedSvi.starts, _hip.ResidentSgd.step = run, starts, step
    try:
        np.random.seed(3)
        glm.fit(X, y, likelihood_args=largs)
    finally:
        _hip.FusedSvi.run, _hip.FusedSvi.starts, _hip.ResidentSgd.step = real_run, real_starts, real_step
        if block is not None:
            glm_mod._FusedLoop.BLOCK_STEPS = old
    flat = lambda v: np.concatenate([np.ravel(np.asarray(u, dtype=float)) for u in (v if isinstance(v, (list, tuple)) else [v])] + [np.empty(0)])  # noqa: E731
    return (glm.weights_.copy(), glm.covariance_.copy(), flat(glm.regularizer_), flat(glm.like_hypers_), flat(glm.basis_hypers_),
            glm.random_.randn()), calls


def _same(a, b, tol):
    for u, v in zip(a[:5], b[:5]):
        assert u.shape == v.shape
        if u.size:
            assert normwise(u, v) < tol, (normwise(u, v), tol)
    assert a[5] == b[5]


@pytest.mark.parametrize("lik", ["poisson", "poisson_softplus", "bernoulli", "binomial", "gaussian"])
def test_fused_loop_equals_the_other_two(lik):
    """The reference's stream (sampler="host"): same minibatches, candidates and draws in all three loops."""
    fused, calls = _fit("fused", lik)
    assert calls["steps"] == 25 and calls["starts"] == 3 and calls["resident"] == 0
    res, calls = _fit("resident", lik)
    assert calls["resident"] == 25 and calls["steps"] == 0
    host, _ = _fit("host", lik)
    _same(fused, host, 2e-5)    # (float64 here against the step-per-call kernels' float32 products)
    _same(fused, res, 2e-5)


@pytest.mark.parametrize("basis", ["iso", "bound", "cat", "cat_ind"])
def test_bases_and_bounds(basis):
    """the isotropic length scale's dimension-0 gradient (basis_functions.py:896), a plain Bound run into (sgd.py:404-420),
    concatenations with their own regularisers / length scales / apply_ind columns"""
    for lik in ("gaussian", "poisson"):
        fused, calls = _fit("fused", lik, basis=basis)
        assert calls["steps"] == 25
        host, _ = _fit("host", lik, basis=basis)
        _same(fused, host, 2e-5)


@pytest.mark.parametrize("name", ["SGDUpdater", "AdaDelta", "AdaGrad", "Momentum", "Adam"])
def test_every_updater(name):
    opt = _imports()[2]
    mk = {"SGDUpdater": lambda: opt.SGDUpdater(eta=1e-4), "Momentum": lambda: opt.Momentum(rho=0.5, eta=1e-4),
          "AdaGrad": lambda: opt.AdaGrad(eta=1e-2)}.get(name, getattr(opt, name))
    fused, _ = _fit("fused", updater=mk, maxiter=12)
    host, _ = _fit("host", updater=mk, maxiter=12)
    _same(fused, host, 2e-5)


def test_device_sampler_equals_the_step_per_call_loop():
    """sampler="device": the kernel's generator IS rr_glm_draw_kernel's function of (seed, step, sample, feature), and the
    random starts consume the step keys the sequential evaluations consume: the two device loops see the same draws."""
    for lik, basis in (("poisson", "ard"), ("gaussian", "cat")):
        fused, calls = _fit("fused", lik, basis=basis, sampler="device")
        assert calls["steps"] == 25 and calls["starts"] == 3
        res, _ = _fit("resident", lik, basis=basis, sampler="device")
        _same(fused, res, 2e-5)


def test_batched_random_starts_score_like_the_sequential_evaluation():
    """500 candidates as one launch: the objective values of the host loop's `_elbo(objective_only)` calls, in order."""
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    rec = []
    fused, calls = _fit("fused", "gaussian", basis="cat", nstarts=40, maxiter=2, record=rec)
    assert calls["starts"] == 40 and len(rec) == 1
    seq = []
    real = GLM._elbo

    def spy(self, *a, objective_only=False, **k):
        out = real(self, *a, objective_only=objective_only, **k)
        if objective_only:
            seq.append(out)
        return out
    GLM._elbo = spy
    try:
        host, _ = _fit("host", "gaussian", basis="cat", nstarts=40, maxiter=2)
    finally:
        GLM._elbo = real
    assert len(seq) == 40
    assert normwise(rec[0], np.array(seq)) < 1e-6
    assert int(np.argmin(rec[0])) == int(np.argmin(seq))
    _same(fused, host, 2e-5)


def test_launch_blocks_and_reruns_are_bit_identical():
    """60 steps in one launch, in launches of 7, and once more: the same bits (fixed-order sums, no atomics)."""
    one, c1 = _fit("fused", "gaussian", basis="cat", maxiter=60, sampler="device")
    cut, c2 = _fit("fused", "gaussian", basis="cat", maxiter=60, sampler="device", block=7)
    again, _ = _fit("fused", "gaussian", basis="cat", maxiter=60, sampler="device", block=7)
    assert c1["run"] == 1 and c2["run"] == 9 and c2["steps"] == 60
    for u, v, w in zip(one[:5], cut[:5], again[:5]):
        assert np.array_equal(u, v) and np.array_equal(v, w)
    one_h, _ = _fit("fused", "poisson", maxiter=30)
    cut_h, c = _fit("fused", "poisson", maxiter=30, block=4)
    assert c["run"] == 8
    for u, v in zip(one_h[:5], cut_h[:5]):
        assert np.array_equal(u, v)
    assert one_h[5] == cut_h[5]


def test_log_lines_show_the_objective_and_parameters_of_their_step(caplog):
    """`Iter n: ELBO = ...` every 500 iterations and at the last (glm.py:232-236, 287-290): same values as the host loop's."""
    with caplog.at_level(logging.INFO, logger="revrand_amd.glm"):
        _fit("fused", "gaussian", basis="cat", maxiter=30, nstarts=0)
        fused_lines = [r.getMessage() for r in caplog.records if r.getMessage().startswith("Iter ")]
        caplog.clear()
        _fit("host", "gaussian", basis="cat", maxiter=30, nstarts=0)
        host_lines = [r.getMessage() for r in caplog.records if r.getMessage().startswith("Iter ")]
    assert len(fused_lines) == len(host_lines) == 2 and fused_lines[0].startswith("Iter 0:") and fused_lines[1].startswith("Iter 29:")

    def numbers(line):
        import re
        return np.array([float(v) for v in re.findall(r"-?\d+\.\d+(?:e[-+]?\d+)?", line)])
    for a, b in zip(fused_lines, host_lines):
        assert normwise(numbers(a), numbers(b)) < 1e-5


def test_shapes_outside_the_fused_range_take_the_step_per_call_loop():
    """minibatch x F beyond one CU's LDS: rr_glm_svi_supported says no and nothing changes for those fits"""
    from revrand_amd import _hip
    assert _hip.svi_supported(83, 10, 50, 10, 3, 6, 2)            # the reference's model-test shape (tests/test_models.py:97-99)
    assert not _hip.svi_supported(2048, 10, 50, 65536, 1, 32, 32)  # config 5
    assert not _hip.svi_supported(83, 64, 50, 10, 3, 6, 2)         # K > 32
    fit, calls = _fit("fused", "poisson", batch=400, maxiter=4)    # 400 rows x 24 features > 8192
    assert calls["steps"] == 0 and calls["resident"] == 4


@pytest.mark.parametrize("shape", [dict(nbases=64, d=8, K=8, L=20, batch=16), dict(nbases=20, d=3, K=1, L=5, batch=7),
                                   dict(nbases=10, d=2, K=32, L=3, batch=1), dict(nbases=33, d=5, K=5, L=1, batch=64),
                                   dict(nbases=24, d=16, K=6, L=17, batch=33)])
def test_shapes_across_the_tiles_of_the_matrix_core_products(shape):
    """The fused loop's two sample products are 16 x 16 x 4 MFMA tiles (samples x rows, rows x features + 1): shapes with
    several tiles per side, ragged last tiles, one sample / one row / one component, K = 32 workgroups, ARD over 16 inputs --
    against the host loop (and rr_glm_svi_supported must accept them)."""
    bs, lk, opt, Bound, Parameter, Positive, GLM = _imports()
    from revrand_amd import _hip
    d, n = shape["d"], shape["nbases"]
    assert _hip.svi_supported(2 * n, shape["K"], shape["L"], shape["batch"], 1, d, d)
    rs = np.random.RandomState(9)
    X = rs.randn(700, d)
    y = rs.poisson(np.exp(0.4 * np.sin(X[:, 0]) + 0.1 * X[:, -1])).astype(float)
    out = []
    for loop in ("fused", "host"):
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
        glm = GLM(lk.Poisson(), basis, K=shape["K"], nsamples=shape["L"], batch_size=shape["batch"], maxiter=15, nstarts=2,
                  random_state=4)
        glm._resident_sgd = loop != "host"
        steps = [0]
        real = _hip.FusedSvi.run

        def run(self, nn, *a, **k):
            steps[0] += nn
            return real(self, nn, *a, **k)
        _hip.FusedSvi.run = run
        try:
            np.random.seed(2)
            glm.fit(X, y)
        finally:
            _hip.FusedSvi.run = real
        assert steps[0] == (15 if loop == "fused" else 0)
        out.append((glm.weights_.copy(), glm.covariance_.copy(), np.atleast_1d(np.asarray(glm.regularizer_, dtype=float)), np.zeros(0),
                    np.ravel(np.asarray(glm.basis_hypers_, dtype=float)), glm.random_.randn()))
    _same(out[0], out[1], 5e-5)
