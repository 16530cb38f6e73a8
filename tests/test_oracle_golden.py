"""
The oracle (oracle/revrand_oracle.py) against the golden vectors that
oracle/make_golden.py produced by importing the reference -- this is what pins
the oracle (tier rule 3).  CPU only.
"""
import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise

SAMPLERS = {
    "RandomRBF": lambda d, n, s: orc.weights_rbf(d, n, s),
    "RandomLaplace": lambda d, n, s: orc.weights_laplace(d, n, s),
    "RandomCauchy": lambda d, n, s: orc.weights_cauchy(d, n, s),
    "RandomMatern32": lambda d, n, s: orc.weights_matern(d, n, s, 1),
    "RandomMatern52": lambda d, n, s: orc.weights_matern(d, n, s, 2),
    "OrthogonalRBF": lambda d, n, s: orc.weights_orthogonal(d, n, s),
}


@pytest.mark.parametrize("cname", sorted(SAMPLERS))
def test_weight_sampling_order(golden, cname):
    g = golden("weights")
    for (d, n, seed) in [(3, 4, 7), (8, 256, 1)]:
        W = g["%s_d%d_n%d_s%d" % (cname, d, n, seed)]
        assert normwise(SAMPLERS[cname](d, n, seed), W) < 1e-14


@pytest.mark.parametrize("cname", sorted(SAMPLERS))
def test_rff_transform_and_grad(golden, cname):
    g = golden("rff")
    for d in ((1, 5, 8) if cname == "RandomRBF" else (5,)):
        X, W = g["X_d%d" % d], g["%s_d%d_W" % (cname, d)]
        for tag, ls in [("iso0.7", 0.7), ("iso2.0", 2.0), ("ard", np.linspace(0.5, 2.0, d))]:
            P = g["%s_d%d_%s_Phi" % (cname, d, tag)]
            dP = g["%s_d%d_%s_dPhi" % (cname, d, tag)]
            assert normwise(orc.rff_transform(X, W, ls), P) < 1e-13
            assert normwise(orc.rff_grad(X, W, ls), dP) < 1e-13
            # shape/index contract: cos block first, (N,2n[,d])
            assert P.shape == (X.shape[0], 2 * W.shape[1])
            assert dP.shape == (P.shape if (tag != "ard" or d == 1) else P.shape + (d,))


def test_iso_grad_quirk_is_dimension_zero_only(golden):
    """Scalar lenscale with d>1: the reference differentiates through X[:,0] only."""
    g = golden("rff")
    X, W = g["X_d5"], g["RandomRBF_d5_W"]
    dP = g["RandomRBF_d5_iso0.7_dPhi"]
    X0 = np.zeros_like(X)
    X0[:, 0] = X[:, 0]
    W0 = np.zeros_like(W)
    W0[0] = W[0]
    Z = X @ (W / 0.7)
    n = W.shape[1]
    dZ = -(X0 @ W0) / 0.7 ** 2
    ref = np.hstack((-np.sin(Z) * dZ, np.cos(Z) * dZ)) / np.sqrt(n)
    assert normwise(ref, dP) < 1e-13


def test_hadamard(golden):
    g = golden("hadamard")
    assert np.array_equal(orc.hadamard(g["doctest_in"], False), g["doctest_nat"])
    assert np.array_equal(orc.hadamard(g["doctest_in"], True), g["doctest_seq"])
    assert normwise(orc.hadamard(g["Y"], False), g["nat"]) < 1e-14
    assert normwise(orc.hadamard(g["Y"], True), g["seq"]) < 1e-14
    for L in (1, 2, 64, 128):
        assert normwise(orc.hadamard(g["Y%d" % L], False), g["nat%d" % L]) < 1e-14


@pytest.mark.parametrize("case", [(1, 10), (2, 10), (5, 16), (16, 64), (128, 256)])
def test_fastfood(golden, case):
    d, nb = case
    g = golden("fastfood")
    k = "d%d_nb%d" % (d, nb)
    B, G, PI, S = orc.fastfood_matrices(nb, d, 3)
    assert np.array_equal(B, g[k + "_B"]) and np.array_equal(PI, g[k + "_PI"])
    assert normwise(G, g[k + "_G"]) < 1e-15 and normwise(S, g[k + "_S"]) < 1e-13
    X = g[k + "_X"]
    assert normwise(orc.fastfood_VX(X, B, G, PI, S), g[k + "_VX"]) < 1e-13
    for tag, ls in [("iso0.7", 0.7), ("iso2.0", 2.0)]:
        assert normwise(orc.fastfood_transform(X, B, G, PI, S, ls), g["%s_%s_Phi" % (k, tag)]) < 1e-12
        assert normwise(orc.fastfood_grad(X, B, G, PI, S, ls), g["%s_%s_dPhi" % (k, tag)]) < 1e-12
    if k + "_ard_Phi" in g:
        ls = np.linspace(0.5, 2.0, d)
        assert normwise(orc.fastfood_transform(X, B, G, PI, S, ls), g[k + "_ard_Phi"]) < 1e-12
        assert normwise(orc.fastfood_grad(X, B, G, PI, S, ls), g[k + "_ard_dPhi"]) < 1e-12


def test_solve_posdef(golden):
    g = golden("solve_posdef")
    for tag in ("pd", "npd"):
        X, ld = orc.solve_posdef(g[tag + "_A"], np.eye(5))
        assert normwise(X, g[tag + "_X"]) < 1e-9
        if np.isfinite(g[tag + "_logdet"]):
            assert abs(ld - g[tag + "_logdet"]) < 1e-9 * max(1, abs(ld))


@pytest.mark.parametrize("tag", ["iso", "ard"])
def test_elbo_single_basis(golden, tag):
    g = golden("elbo")
    X, y, W, ls = g["X"], g["y"], g[tag + "_W"], g[tag + "_ls"]
    n = W.shape[1]
    Phi = orc.rff_transform(X, W, ls)
    dP = orc.rff_grad(X, W, ls)
    dPl = [dP] if dP.ndim == 2 else [dP[:, :, i] for i in range(dP.shape[2])]
    o = orc.slm_elbo(Phi, y, float(g["var"]), np.full(2 * n, float(g["reg"])), slice(None), dPl)
    assert abs(o["elbo"] - g[tag + "_elbo"]) < 1e-10 * abs(g[tag + "_elbo"])
    assert normwise(o["m"], g[tag + "_m"]) < 1e-9
    assert normwise(o["C"], g[tag + "_C"]) < 1e-9
    assert normwise(o["dvar"], g[tag + "_dvar"]) < 1e-9
    assert normwise(o["dreg"], g[tag + "_dreg"]) < 1e-9
    assert normwise(np.array(o["dhyp"]), g[tag + "_dhyp"]) < 1e-8
    # the sufficient-statistics route gives the same posterior (SURVEY a-12)
    m, C, ldC = orc.slm_posterior_from_stats(g[tag + "_G"], g[tag + "_b"], float(g["var"]),
                                            np.full(2 * n, float(g["reg"])))
    assert normwise(m, g[tag + "_m"]) < 1e-9 and abs(ldC - g[tag + "_logdetC"]) < 1e-8


def test_fit_prediction_moments(golden):
    g = golden("fit")
    Phi = orc.rff_transform(g["Xs"], g["W"], float(g["hyp_"]))
    Ey, Vy = orc.slm_predict_moments(Phi, g["m"], g["C"], float(g["var_"]))
    assert normwise(Ey, g["Ey"]) < 1e-12 and normwise(Vy, g["Vy"]) < 1e-12


@pytest.mark.parametrize("case", [(1, 10, 3), (3, 8, 4), (5, 16, 3), (16, 32, 3)])
def test_fastfood_gm(golden, case):
    d, nb, seed = case
    g = golden("fastfood_gm")
    k = "d%d_nb%d" % (d, nb)
    B, G, PI, S = orc.fastfood_matrices(nb, d, seed)
    X, mean, ls = g[k + "_X"], g[k + "_mean"], g[k + "_ls"]
    assert normwise(orc.fastfood_gm_transform(X, B, G, PI, S, mean, ls), g[k + "_Phi"]) < 1e-12
    dM, dL = orc.fastfood_gm_grad(X, B, G, PI, S, mean, ls)
    assert normwise(dM, g[k + "_dmean"]) < 1e-12 and normwise(dL, g[k + "_dlen"]) < 1e-12


GLM_CASES = [("iso", n) for n in ("poisson_exp", "poisson_softplus", "gaussian", "bernoulli", "binomial")] \
    + [("ard", "poisson_exp"), ("ard", "gaussian")]


@pytest.mark.parametrize("tag,lik", GLM_CASES)
def test_glm_minibatch_elbo(golden, tag, lik):
    """One SVI minibatch of the reference's GLM `_elbo` (glm.py:205-322), seeded: -ELBO and the five
    gradient blocks, with the reference's own standard-normal draws."""
    g = golden("glm")
    X, W, ls = g["X"], g["W"], g[tag + "_ls"]
    ls = float(ls) if np.ndim(ls) == 0 else ls
    d, D = X.shape[1], 2 * W.shape[1]
    t = tag + "_" + lik
    Phi = orc.rff_transform(X, W, ls)
    dP = orc.rff_grad(X, W, ls)
    dPs = [dP[:, :, i] for i in range(d)] if tag == "ard" else [dP]
    lpars = [0.7] if lik == "gaussian" else []
    largs = (g["nbin"],) if lik == "binomial" else ()
    nobj, (ndm, ndC, dL, dlp, dbp) = orc.glm_elbo(g["m"], g["C"], np.full(D, float(g["reg"])), slice(None), lik, lpars,
                                                  largs, Phi, dPs, g[t + "_y"], g["e"], float(g["B"]))
    assert abs(nobj - g[t + "_obj"]) < 1e-10 * abs(g[t + "_obj"])
    assert normwise(ndm, g[t + "_ndm"]) < 1e-10 and normwise(ndC, g[t + "_ndC"]) < 1e-10
    assert abs(dL[0] - g[t + "_dL"]) < 1e-10 * abs(g[t + "_dL"])
    assert normwise(np.array(dbp), g[t + "_dbp"]) < 1e-10
    if lpars:
        assert normwise(np.atleast_1d(dlp[0]), g[t + "_dlp"]) < 1e-10


@pytest.mark.parametrize("name", ["sgd", "adadelta", "adagrad", "momentum", "adam"])
def test_sgd_updaters(golden, name):
    g = golden("glm")
    x, st = np.linspace(-1, 1, 6), {}
    for i, grad in enumerate(g["upd_grads"]):
        x = orc.sgd_update(name, st, x, grad)
        assert normwise(x, g["upd_" + name][i]) < 1e-12


def test_concat_transform_grad_regulariser(golden):
    """BasisCat (RandomMatern52 + LinearBasis): hstack'ed Phi, the zero-padded gradient, the per-basis regulariser
    diagonal + slices and get_dim as the REFERENCE returned them (basis_functions.py:1599-1748)."""
    g = golden("concat")
    X, ls, W = g["X"], g["ls"], g["W"]
    N, d = X.shape
    n = W.shape[1]
    assert np.array_equal(orc.weights_matern(d, n, 4, 2), W)
    Phi = np.hstack((orc.rff_transform(X, W, ls), orc.linear_transform(X, True)))
    assert normwise(Phi, g["Phi"]) < 1e-12 and int(g["get_dim"]) == Phi.shape[1] == 2 * n + d + 1
    dPhi = np.zeros((N, Phi.shape[1], d))
    dPhi[:, :2 * n, :] = orc.rff_grad(X, W, ls)
    assert normwise(dPhi, g["dPhi"]) < 1e-12
    assert np.all(g["dPhi"][:, 2 * n:, :] == 0)  # the linear block has no hyper-parameter: exact zeros
    assert g["slices"].tolist() == [[0, 2 * n], [2 * n, 2 * n + d + 1]]
    assert np.array_equal(g["regdiag"], np.concatenate((np.full(2 * n, 2.5), np.full(d + 1, 0.5))))


def test_concat_apply_ind(golden):
    """The apply_ind case modelled on the reference's tests/test_bases.py:223-238: every basis sees its own column
    subset (basis_functions.py:70-105), in the order the index list gives."""
    g = golden("concat")
    X, W1, W2 = g["ai_X"], g["ai_W1"], g["ai_W2"]
    assert np.array_equal(orc.weights_rbf(1, 1, 8), W1) and np.array_equal(orc.weights_rbf(2, 3, 9), W2)
    ls2 = np.array([0.8, 1.7])
    Phi = np.hstack((X[:, [0]], orc.rff_transform(X[:, [1]], W1, 1.5), orc.rff_transform(X[:, [1, 0]], W2, ls2)))
    assert normwise(Phi, g["ai_Phi"]) < 1e-12
    g0 = np.zeros((X.shape[0], 9))
    g0[:, 1:3] = orc.rff_grad(X[:, [1]], W1, 1.5)
    g1 = np.zeros((X.shape[0], 9, 2))
    g1[:, 3:9, :] = orc.rff_grad(X[:, [1, 0]], W2, ls2)
    assert normwise(g0, g["ai_dPhi0"]) < 1e-12 and normwise(g1, g["ai_dPhi1"]) < 1e-12


def c1_data(N=10000, d=8, seed=11):
    """BASELINE config 1's data set, regenerated from the seed exactly as oracle/make_golden.py does."""
    r = np.random.RandomState(seed)
    X = r.randn(N, d)
    w = np.array([1.0, -0.7, 0.5, 0.3, -0.2, 0.9, -0.4, 0.1])[:d]
    y = np.sin(X @ w) + 0.1 * r.randn(N)
    Xs = np.random.RandomState(seed + 1).randn(64, d)
    return X, y, Xs


def test_fit_c1_shape_prediction_mean(golden):
    """Config 1 at its real shape (RandomRBF nbases=256, D=8, N=10k, fit by the reference): the predictive mean follows
    from the stored posterior weights through the oracle's transform, and the fit is a fit (SMSE on noise-free targets)."""
    g = golden("fit_c1")
    _, _, Xs = c1_data()
    W = orc.weights_rbf(8, 256, 41)
    assert np.array_equal(W[:, :8], g["c1_W_head"])
    Phi = orc.rff_transform(Xs, W, g["c1_hyp_"])
    assert normwise(Phi @ g["c1_m"], g["c1_Ey"]) < 1e-10
    smse = ((g["c1_ys_true"] - g["c1_Ey"]) ** 2).sum() / (64 * g["c1_ys_true"].var())
    assert smse < 0.3 and np.all(g["c1_Vy"] > float(g["c1_var_"]))  # 256 bases in 8 dimensions: a coarse fit
    # the reference's _elbo at its fitted point (N = 10k, F = 512): the oracle reproduces objective, gradients, posterior
    X, y, _ = c1_data()
    ls = float(g["c1_hyp_"])
    o = orc.slm_elbo(orc.rff_transform(X, W, ls), y, float(g["c1_var_"]), np.full(512, float(g["c1_reg_"])), slice(None),
                     [orc.rff_grad(X, W, ls)])
    assert abs(o["elbo"] - float(g["c1_at_elbo"])) < 1e-10 * abs(float(g["c1_at_elbo"]))
    assert normwise(o["m"], g["c1_at_m"]) < 1e-9 and normwise(o["C"].diagonal(), g["c1_at_Cdiag"]) < 1e-9
    assert abs(o["dvar"] - float(g["c1_at_dvar"])) < 1e-8 * abs(float(g["c1_at_dvar"]))
    assert normwise(np.atleast_1d(o["dhyp"]), g["c1_at_dhyp"]) < 1e-7
    # second small fit (ARD Matern32): same consistency
    Phi2 = orc.rff_transform(g["s2_Xs"], g["s2_W"], g["s2_hyp_"])
    assert np.array_equal(orc.weights_matern(4, 20, 43, 1), g["s2_W"]) and Phi2.shape == (16, 40)


def test_converged_fit_fixture_is_a_stationary_point_of_the_oracles_elbo(golden):
    """tests/golden/fit_converged.npz (a reference fit that CONVERGED: config 1's shape, ARD length scales): the oracle's
    ELBO at the stored optimum equals the reference's objective, its posterior mean the reference's weights, and the point
    is stationary -- the gradient in log-space (what L-BFGS-B sees through the log trick) is below 1e-4 of the objective
    there."""
    g = golden("fit_converged")
    X, y, Xs = c1_data()
    W = orc.weights_rbf(8, 256, 41)
    assert np.array_equal(W[:, :8], g["W_head"]) and int(g["nit"]) < 100
    ls, var, reg = g["hyp_"], float(g["var_"]), float(g["reg_"])
    dP = orc.rff_grad(X, W, ls)
    o = orc.slm_elbo(orc.rff_transform(X, W, ls), y, var, np.full(512, reg), slice(None), [dP[:, :, i] for i in range(8)])
    assert abs(o["elbo"] - float(g["obj"])) < 1e-9 * abs(float(g["obj"]))
    assert normwise(o["m"], g["m"]) < 1e-8 and normwise(o["C"].diagonal(), g["Cdiag"]) < 1e-8
    glog = np.concatenate(([o["dvar"] * var], [o["dreg"][0] * reg], np.asarray(o["dhyp"]) * ls))
    assert np.abs(glog).max() < 1e-4 * abs(float(g["obj"])), glog   # (0.15 against an objective of 5448; L-BFGS-B stopped on ftol)
    Phi = orc.rff_transform(Xs, W, ls)
    assert normwise(Phi @ g["m"], g["Ey"]) < 1e-10
    assert abs(float(g["smse"]) - ((g["ys_true"] - g["Ey"]) ** 2).mean() / g["ys_true"].var()) < 1e-12 and float(g["smse"]) < 0.03


def test_oracle_gradient_trace_identity_above_its_size_switch():
    """oracle.slm_elbo forms sum((dPhi^T Phi) o C) -- slm.py:193-195 as written -- up to F = 512 and through the identity
    sum(dPhi o (Phi C)) above (one product instead of one per length scale): the two are the same number."""
    rs = np.random.RandomState(3)
    N, F, d = 300, 600, 3
    Phi = rs.randn(N, F) / np.sqrt(F)
    y = rs.randn(N)
    dPs = [rs.randn(N, F) / np.sqrt(F) for _ in range(d)]
    o = orc.slm_elbo(Phi, y, 0.4, np.full(F, 1.3), slice(None), dPs)
    err = y - Phi @ o["m"]
    want = [(o["m"] @ (err @ dP) - ((dP.T @ Phi) * o["C"]).sum()) / 0.4 for dP in dPs]
    assert np.allclose(o["dhyp"], want, rtol=1e-11, atol=1e-12)


from glm_fit_cases import CASES as GLM_FIT_CASES, updater_of  # noqa: E402


@pytest.mark.parametrize("case", GLM_FIT_CASES, ids=[c[0] for c in GLM_FIT_CASES])
def test_glm_fit_optimiser_stack(golden, case):
    """`orc.glm_fit` -- random starts, structured_sgd / logtrick_sgd, sgd with bounds, one RandomState shared by the
    permutations and the reparameterisation draws -- against `GeneralizedLinearModel.fit` of the reference: fitted blocks,
    the optimiser's gradient norms and objectives, and the stream's end state."""
    from scipy.stats import gamma
    g = golden("glm_fit")
    tag, lik, kind, bs, ns, fwd = case
    d, n = g["X"].shape[1], int(g["nbases"])
    P = orc.ParamSpec
    reg = lambda: P(dist=gamma(1.), positive=True)  # noqa: E731  (basis_functions.py:210: the default regulariser)
    if kind == "cat":
        ch = [("linear", True), ("rff", g[tag + "_W1"], 1), ("rff", g[tag + "_W2"], 1)]
        regs, lss = [reg(), reg(), reg()], [P(value=[]), P(dist=gamma(1.), positive=True), P(dist=gamma(1.), positive=True)]
    else:
        ls = {"ard": P(dist=gamma(4., scale=0.25), positive=True, shape=(d,)), "bound": P(value=1.0, lower=0.996, upper=1.001),
              "posupper": P(value=1.0, positive=True, upper=1.03)}[kind]
        ch, regs, lss = [("rff", g[tag + "_W0"], d if kind == "ard" else 1)], [reg()], [ls]
    likpar = [P(dist=gamma(1.), positive=True)] if lik == "gaussian" else []
    largs = [g["nbin"]] if lik == "binomial" else []
    o = orc.glm_fit(g["X"], g["y_" + ("poisson_exp" if lik == "poisson_softplus" else lik)], lik, largs, ch, regs, likpar, lss, int(g["K"]), int(g["L"]), bs, int(g["maxiter"]),
                    ns, int(g["seed"]), int(g["global_seed"]), sgd_batch_size=bs if fwd else 10,
                    **({} if updater_of(tag) is None else {"updater": updater_of(tag)[2], "updater_hp": updater_of(tag)[1]}))
    flat = lambda v: np.concatenate([np.ravel(np.asarray(u, float)) for u in v] + [np.empty(0)])  # noqa: E731
    assert normwise(o[0], g[tag + "_m"]) < 1e-9 and normwise(o[1], g[tag + "_C"]) < 1e-9
    assert normwise(flat(o[2]), g[tag + "_reg"]) < 1e-9
    assert normwise(flat(o[3]), g[tag + "_lik"]) < 1e-9
    assert normwise(flat(o[4]), g[tag + "_ls"]) < 1e-9
    assert normwise(o[6], g[tag + "_norms"]) < 1e-9
    fin = np.isfinite(g[tag + "_objs"])
    assert np.array_equal(fin, np.isfinite(o[5])) and normwise(o[5][fin], g[tag + "_objs"][fin]) < 1e-9
    assert o[7] == float(g[tag + "_end"])
    if kind == "bound":   # the case is there for the bounds: a step of Adam's 0.01 from 1.0 leaves [0.996, 1.001] at once
        assert 0.996 <= float(o[4][0]) <= 1.001
    if kind == "posupper":
        assert float(o[4][0]) == pytest.approx(1.03, abs=1e-12)


@pytest.mark.parametrize("lik", ["gaussian", "poisson_exp", "binomial"])
def test_glm_predictions(golden, lik):
    """`orc.glm_predictions` against the reference's `_sample_func` / `predict_moments` / `predict_logpdf` / `predict_cdf`
    (glm.py:351-495, 572-620) on the reference's own draws."""
    g = golden("glm_predict")
    largs = [g["nbin"]] if lik == "binomial" else []
    lhyp = [0.3] if lik == "gaussian" else []
    o = orc.glm_predictions(g["X"], g["m"], g["C"], [("linear", True), ("rff", g["W"], 1)], [[], float(g["ls"])], lik, lhyp, largs,
                            int(g["S"]), 77, g["yq_" + lik], float(g[lik + "_q"]))
    assert normwise(o["fs"], g[lik + "_fs"]) < 1e-12
    assert normwise(o["Ey"], g[lik + "_Ey"]) < 1e-12 and normwise(o["Vy"], g[lik + "_Vy"]) < 1e-12
    assert normwise(np.array(o["logpdf"]), g[lik + "_logpdf"]) < 1e-12 and normwise(np.array(o["cdf"]), g[lik + "_cdf"]) < 1e-12


@pytest.mark.parametrize("tag,ns", [("ns6", 6), ("ns1", 1)])
def test_slm_random_starts(golden, tag, ns):
    """`orc.slm_random_starts` against the start the reference's `fit` hands to L-BFGS-B (maxiter = 0 returns it), every
    candidate's objective and the stream's end state."""
    from scipy.stats import gamma
    g = golden("slm_starts")
    P = orc.ParamSpec
    d = g["X"].shape[1]
    (var, reg, ls), objs, end = orc.slm_random_starts(g["X"], g["y"], g["W"], P(dist=gamma(1.), positive=True), P(dist=gamma(1.), positive=True),
                                                      P(dist=gamma(2., scale=0.5), positive=True, shape=(d,)), ns, 13)
    assert normwise(objs, g[tag + "_cand_objs"]) < 1e-10 and end == float(g[tag + "_end"])
    assert abs(var - float(g[tag + "_var"])) < 1e-12 and abs(reg - float(g[tag + "_reg"])) < 1e-12
    assert normwise(ls, g[tag + "_hyp"]) < 1e-12
