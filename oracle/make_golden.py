#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE
(/root/reference, revrand v1.0.0) in the build container, and check the NumPy
oracle (oracle/revrand_oracle.py) against every one of them while doing so.

Run (build container only -- /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python -B oracle/make_golden.py

Only *data* (inputs + the reference's outputs) is written; no reference source
travels.  Two process-local accommodations are needed to import the 2017-era
reference under NumPy 2: the ``decorator`` stand-in in oracle/shim
and ``np.asscalar`` (removed in NumPy 2.0, used at revrand/utils/base.py:285).
Neither touches /root/reference.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

if not hasattr(np, "asscalar"):
    np.asscalar = lambda a: a.item()  # process-local, see docstring

import revrand.basis_functions as rb  # noqa: E402
from revrand.btypes import Parameter, Positive  # noqa: E402
from revrand.mathfun.linalg import hadamard as ref_hadamard, solve_posdef as ref_solve  # noqa: E402
from revrand.slm import StandardLinearModel  # noqa: E402

import revrand_oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

RFF = {
    "RandomRBF": (rb.RandomRBF, lambda d, n, s: orc.weights_rbf(d, n, s)),
    "RandomLaplace": (rb.RandomLaplace, lambda d, n, s: orc.weights_laplace(d, n, s)),
    "RandomCauchy": (rb.RandomCauchy, lambda d, n, s: orc.weights_cauchy(d, n, s)),
    "RandomMatern32": (rb.RandomMatern32, lambda d, n, s: orc.weights_matern(d, n, s, 1)),
    "RandomMatern52": (rb.RandomMatern52, lambda d, n, s: orc.weights_matern(d, n, s, 2)),
    "OrthogonalRBF": (rb.OrthogonalRBF, lambda d, n, s: orc.weights_orthogonal(d, n, s)),
}


def close(a, b, tol=1e-12):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, np.abs(a).max() if a.size else 1.0)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol * scale, err
    return err


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB  %d arrays" % (name, os.path.getsize(path) / 1024., len(arrs)))


def gen_weights():
    out = {}
    for cname, (cls, ofun) in RFF.items():
        for (d, n, seed) in [(3, 4, 7), (8, 256, 1)]:
            W = cls(nbases=n, Xdim=d, random_state=seed).W
            assert np.array_equal(W.shape, (d, n))
            close(W, ofun(d, n, seed), 1e-14)
            out["%s_d%d_n%d_s%d" % (cname, d, n, seed)] = W
    save("weights", **out)


def gen_rff():
    out = {}
    N, n = 24, 12
    for cname, (cls, ofun) in RFF.items():
        dims = (1, 5, 8) if cname == "RandomRBF" else (5,)
        for d in dims:
            X = np.random.RandomState(0).randn(N, d)
            out["X_d%d" % d] = X
            seed = 11 + d
            key = "%s_d%d" % (cname, d)
            for tag, ls in [("iso0.7", 0.7), ("iso2.0", 2.0),
                            ("ard", np.linspace(0.5, 2.0, d))]:
                if tag == "ard":
                    b = cls(nbases=n, Xdim=d, random_state=seed,
                            lenscale=Parameter(np.ones(d), Positive()))
                else:
                    b = cls(nbases=n, Xdim=d, random_state=seed)
                out[key + "_W"] = b.W
                P = b.transform(X, ls)
                dP = b.grad(X, ls)
                assert P.dtype == np.float64
                close(P, orc.rff_transform(X, b.W, ls))
                close(dP, orc.rff_grad(X, b.W, ls))
                out["%s_%s_Phi" % (key, tag)] = P
                out["%s_%s_dPhi" % (key, tag)] = dP
            out[key + "_seed"] = np.array(seed)
    # float32 inputs: output stays float64 (np.dot promotes), SURVEY a-1
    X32 = np.random.RandomState(0).randn(N, 5).astype(np.float32)
    b = rb.RandomRBF(nbases=n, Xdim=5, random_state=16)
    P = b.transform(X32, 1.3)
    assert P.dtype == np.float64
    out["RandomRBF_d5_f32in_iso1.3_Phi"] = P
    close(P, orc.rff_transform(X32, b.W, 1.3))
    save("rff", **out)


def gen_hadamard():
    Y = np.random.RandomState(5).randn(3, 16)
    out = dict(Y=Y, nat=ref_hadamard(Y, ordering=False), seq=ref_hadamard(Y, ordering=True))
    close(out["nat"], orc.hadamard(Y, False))
    close(out["seq"], orc.hadamard(Y, True))
    # the reference's own doctest vector (mathfun/linalg.py:202-206)
    y = np.array([[1, 0, 1, 0, 0, 1, 1, 0]])
    out["doctest_in"] = y
    out["doctest_nat"] = ref_hadamard(y, ordering=False)
    out["doctest_seq"] = ref_hadamard(y, ordering=True)
    assert np.array_equal(out["doctest_nat"], [[.5, .25, 0, -.25, 0, .25, 0, .25]])
    assert np.array_equal(out["doctest_seq"], [[.5, 0, 0, 0, -.25, .25, .25, .25]])
    for L in (1, 2, 64, 128):
        Z = np.random.RandomState(L).randn(2, L)
        out["Y%d" % L] = Z
        out["nat%d" % L] = ref_hadamard(Z, ordering=False)
        close(out["nat%d" % L], orc.hadamard(Z, False))
    save("hadamard", **out)


def gen_fastfood():
    out = {}
    cases = [(1, 10, 3, 6, True), (2, 10, 3, 6, True), (5, 16, 3, 6, True),
             (16, 64, 3, 4, True), (128, 256, 3, 4, False)]
    for (d, nb, seed, N, ard) in cases:
        key = "d%d_nb%d" % (d, nb)
        X = np.random.RandomState(1).randn(N, d)
        b = rb.FastFoodRBF(nbases=nb, Xdim=d, random_state=seed)
        B, G, PI, S = orc.fastfood_matrices(nb, d, seed)
        assert np.array_equal(B, b.B) and np.array_equal(PI, b.PI)
        close(G, b.G, 1e-15)
        close(S, b.S, 1e-13)
        assert orc.fastfood_dims(nb, d) == (b.d2, b.k, b.n)
        out.update({key + "_X": X, key + "_B": b.B, key + "_G": b.G,
                    key + "_PI": b.PI, key + "_S": b.S})
        VX = b._makeVX(X)
        close(VX, orc.fastfood_VX(X, B, G, PI, S))
        out[key + "_VX"] = VX
        for tag, ls in [("iso0.7", 0.7), ("iso2.0", 2.0)]:
            P, dP = b.transform(X, ls), b.grad(X, ls)
            close(P, orc.fastfood_transform(X, B, G, PI, S, ls))
            close(dP, orc.fastfood_grad(X, B, G, PI, S, ls))
            out["%s_%s_Phi" % (key, tag)] = P
            out["%s_%s_dPhi" % (key, tag)] = dP
        if ard:
            ls = np.linspace(0.5, 2.0, d)
            ba = rb.FastFoodRBF(nbases=nb, Xdim=d, random_state=seed,
                                lenscale=Parameter(np.ones(d), Positive()))
            P, dP = ba.transform(X, ls), ba.grad(X, ls)
            close(P, orc.fastfood_transform(X, B, G, PI, S, ls))
            close(dP, orc.fastfood_grad(X, B, G, PI, S, ls))
            out[key + "_ard_Phi"] = P
            out[key + "_ard_dPhi"] = dP
    save("fastfood", **out)


def gen_fastfood_gm():
    from revrand.btypes import Bound
    out = {}
    for (d, nb, seed, N) in [(1, 10, 3, 6), (3, 8, 4, 5), (5, 16, 3, 6), (16, 32, 3, 4)]:
        key = "d%d_nb%d" % (d, nb)
        X = np.random.RandomState(1).randn(N, d)
        mean = np.linspace(-0.7, 0.9, d)
        ls = np.linspace(0.6, 1.8, d)
        b = rb.FastFoodGM(nbases=nb, Xdim=d, random_state=seed, mean=Parameter(np.zeros(d), Bound()),
                          lenscale=Parameter(np.ones(d), Positive()))
        B, G, PI, S = orc.fastfood_matrices(nb, d, seed)
        assert np.array_equal(B, b.B) and np.array_equal(PI, b.PI)
        P = b.transform(X, mean, ls)
        dM, dL = b.grad(X, mean, ls)
        close(P, orc.fastfood_gm_transform(X, B, G, PI, S, mean, ls))
        oM, oL = orc.fastfood_gm_grad(X, B, G, PI, S, mean, ls)
        close(dM, oM)
        close(dL, oL)
        out.update({key + "_X": X, key + "_mean": mean, key + "_ls": ls, key + "_Phi": P, key + "_dmean": dM,
                    key + "_dlen": dL})
    save("fastfood_gm", **out)


def gen_concat():
    out = {}
    d, n, N = 6, 10, 20
    X = np.random.RandomState(2).randn(N, d)
    ls = np.linspace(0.5, 2.0, d)
    base = rb.RandomMatern52(nbases=n, Xdim=d, random_state=4,
                             lenscale=Parameter(np.ones(d), Positive())) \
        + rb.LinearBasis(onescol=True)
    P = base.transform(X, ls)
    grads = list(base.grad(X, ls))
    assert len(grads) == 1 and grads[0].shape == (N, 2 * n + d + 1, d)
    diag, slices = base.regularizer_diagonal(X, 2.5, 0.5)
    W = base.bases[0].W
    Pref = np.hstack((orc.rff_transform(X, W, ls), orc.linear_transform(X)))
    close(P, Pref)
    gref = np.zeros((N, 2 * n + d + 1, d))
    gref[:, :2 * n, :] = orc.rff_grad(X, W, ls)
    close(grads[0], gref)
    out.update(X=X, ls=ls, W=W, Phi=P, dPhi=grads[0], regdiag=diag,
               slices=np.array([[s.start, s.stop] for s in slices]),
               get_dim=np.array(base.get_dim(X)))
    # apply_ind case modelled on tests/test_bases.py:223-238
    X2 = np.random.RandomState(3).randn(N, 2)
    base2 = rb.LinearBasis(onescol=False, apply_ind=[0]) \
        + rb.RandomRBF(Xdim=1, nbases=1, apply_ind=[1], random_state=8) \
        + rb.RandomRBF(Xdim=2, nbases=3, random_state=9,
                       lenscale=Parameter(np.ones(2), Positive()), apply_ind=[1, 0])
    P2 = base2.transform(X2, 1.5, np.array([0.8, 1.7]))
    g2 = list(base2.grad(X2, 1.5, np.array([0.8, 1.7])))
    assert P2.shape == (N, 9) and g2[0].shape == (N, 9) and g2[1].shape == (N, 9, 2)
    out.update(ai_X=X2, ai_W1=base2.bases[1].W, ai_W2=base2.bases[2].W, ai_Phi=P2,
               ai_dPhi0=g2[0], ai_dPhi1=g2[1])
    save("concat", **out)


def _elbo_case(basis, X, y, var, reg, hypers):
    slm = StandardLinearModel(basis)
    slm.obj_ = -np.inf
    nelbo, (ndvar, ndreg, ndhyp) = slm._elbo(X, y, var, reg, hypers)
    return dict(elbo=-nelbo, dvar=-np.asarray(ndvar),
                dreg=-np.atleast_1d(np.asarray(ndreg, float)),
                dhyp=[-np.atleast_1d(np.asarray(h, float)) for h in
                      (ndhyp if isinstance(ndhyp, list) else [ndhyp])],
                m=slm.weights_, C=slm.covariance_)


def gen_elbo():
    out = {}
    N, d, n = 500, 4, 16
    r = np.random.RandomState(6)
    X = r.randn(N, d)
    y = np.sin(X @ r.randn(d)) + 0.1 * r.randn(N)
    out.update(X=X, y=y)
    var = 0.37
    ard = np.linspace(0.6, 1.8, d)

    # single basis, iso (with the dimension-0-only gradient quirk) and ARD
    for tag, ls, lsp in [("iso", 1.3, Parameter(1., Positive())),
                         ("ard", ard, Parameter(np.ones(d), Positive()))]:
        b = rb.RandomRBF(nbases=n, Xdim=d, random_state=21, lenscale=lsp)
        res = _elbo_case(b, X, y, var, 1.7, ls)
        Phi = orc.rff_transform(X, b.W, ls)
        dP = orc.rff_grad(X, b.W, ls)
        dPl = [dP] if dP.ndim == 2 else [dP[:, :, i] for i in range(d)]
        o = orc.slm_elbo(Phi, y, var, np.full(2 * n, 1.7), slice(None), dPl)
        close(res["elbo"], o["elbo"], 1e-11)
        close(res["m"], o["m"], 1e-9)
        close(res["C"], o["C"], 1e-9)
        close(res["dvar"], o["dvar"], 1e-9)
        close(res["dreg"], o["dreg"], 1e-9)
        close(np.concatenate(res["dhyp"]), np.array(o["dhyp"]), 1e-8)
        out.update({tag + "_W": b.W, tag + "_ls": np.asarray(ls), tag + "_elbo": res["elbo"],
                    tag + "_dvar": res["dvar"], tag + "_dreg": res["dreg"],
                    tag + "_dhyp": np.concatenate(res["dhyp"]), tag + "_m": res["m"],
                    tag + "_C": res["C"], tag + "_G": o["G"], tag + "_b": o["b"],
                    tag + "_logdetC": o["logdetC"]})

    # concatenation: Matern52 (ARD) + LinearBasis, two regularisers
    b = rb.RandomMatern52(nbases=n, Xdim=d, random_state=22,
                          lenscale=Parameter(np.ones(d), Positive())) + rb.LinearBasis(onescol=True)
    res = _elbo_case(b, X, y, var, [1.7, 0.6], ard)
    W = b.bases[0].W
    Phi = np.hstack((orc.rff_transform(X, W, ard), orc.linear_transform(X)))
    F = Phi.shape[1]
    dP = np.zeros((N, F, d))
    dP[:, :2 * n, :] = orc.rff_grad(X, W, ard)
    rd = np.concatenate((np.full(2 * n, 1.7), np.full(d + 1, 0.6)))
    o = orc.slm_elbo(Phi, y, var, rd, [slice(0, 2 * n), slice(2 * n, F)],
                     [dP[:, :, i] for i in range(d)])
    close(res["elbo"], o["elbo"], 1e-11)
    close(res["dreg"], o["dreg"], 1e-9)
    close(np.concatenate(res["dhyp"]), np.array(o["dhyp"]), 1e-8)
    out.update(cat_W=W, cat_ls=ard, cat_reg=np.array([1.7, 0.6]), cat_elbo=res["elbo"],
               cat_dvar=res["dvar"], cat_dreg=res["dreg"], cat_dhyp=np.concatenate(res["dhyp"]),
               cat_m=res["m"], cat_C=res["C"])
    out["var"] = np.array(var)
    out["reg"] = np.array(1.7)
    save("elbo", **out)


def gen_solve_posdef():
    r = np.random.RandomState(99)
    Xc = r.randn(100, 5)
    S = np.cov(Xc.T)
    l, U = np.linalg.eigh(S)
    l2 = l.copy()
    l2[0] = -1e-13
    Sn = (U * l2) @ U.T
    out = {}
    for tag, A in [("pd", S), ("npd", Sn)]:
        Xr, ld = ref_solve(A, np.eye(5))
        Xo, lo = orc.solve_posdef(A, np.eye(5))
        close(Xr, Xo, 1e-9)
        if np.isfinite(ld):
            close(ld, lo, 1e-9)
        out.update({tag + "_A": A, tag + "_X": Xr, tag + "_logdet": np.array(ld)})
    save("solve_posdef", **out)


def gen_fit():
    """End-to-end C1-small: fit with fixed inits, nstarts=0, maxiter=20 (a-14)."""
    N, d, n = 400, 3, 24
    r = np.random.RandomState(7)
    X = r.randn(N, d)
    y = np.sin(X @ np.array([1.0, -0.5, 0.3])) + 0.05 * r.randn(N)
    Xs = np.random.RandomState(8).randn(16, d)
    b = rb.RandomRBF(nbases=n, Xdim=d, random_state=31, lenscale=Parameter(1.2, Positive()),
                     regularizer=Parameter(1.5, Positive()))
    slm = StandardLinearModel(b, var=Parameter(0.5, Positive()), nstarts=0, maxiter=20)
    slm.fit(X, y)
    Ey, Vy = slm.predict_moments(Xs)
    save("fit", X=X, y=y, Xs=Xs, W=b.W, var_=np.array(slm.var_), reg_=np.array(slm.regularizer_),
         hyp_=np.array(slm.hypers_), m=slm.weights_, C=slm.covariance_, obj=np.array(slm.obj_),
         Ey=Ey, Vy=Vy)


def c1_data(N=10000, d=8, seed=11):
    """Config 1's synthetic data set (BASELINE.json configs[0]: RandomRBF nbases=256, D=8, N=10k): regenerated from
    the seed by the tests, so only the reference's OUTPUTS are stored."""
    r = np.random.RandomState(seed)
    X = r.randn(N, d)
    w = np.array([1.0, -0.7, 0.5, 0.3, -0.2, 0.9, -0.4, 0.1])[:d]
    y = np.sin(X @ w) + 0.1 * r.randn(N)
    Xs = np.random.RandomState(seed + 1).randn(64, d)
    return X, y, Xs


def gen_fit_c1():
    """BASELINE config 1 at its real shape through the reference's StandardLinearModel.fit (slm.py:74-140): nstarts=0,
    fixed initial values, maxiter=20; and the small fit of gen_fit() again with a second seed / ARD length scales."""
    X, y, Xs = c1_data()
    # start values chosen so that the reference's L-BFGS-B makes progress: from (0.3, 1.0, 1.5) or (1, 1, 1) its first
    # log-space step overshoots to var ~ 1e-100, the line search gives up and fit() returns the start point
    b = rb.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(2.0, Positive()),
                     regularizer=Parameter(10.0, Positive()))
    slm = StandardLinearModel(b, var=Parameter(0.02, Positive()), nstarts=0, maxiter=20)
    slm.fit(X, y)
    assert abs(float(slm.var_) - 0.02) > 1e-3  # the optimiser moved
    Ey, Vy = slm.predict_moments(Xs)
    Phi = orc.rff_transform(Xs, b.W, np.asarray(slm.hypers_))
    Eo, Vo = orc.slm_predict_moments(Phi, slm.weights_, slm.covariance_, float(slm.var_))
    close(Ey, Eo, 1e-10)
    close(Vy, Vo, 1e-10)
    # one evaluation of the reference's _elbo AT its fitted point (weights_ / obj_ above belong to the best point SEEN,
    # which need not be the optimiser's last): objective, all gradients, posterior -- what a tight comparison can use
    fit_obj, fit_m = float(slm.obj_), slm.weights_.copy()
    slm.obj_ = -np.inf
    at = _elbo_case(b, X, y, float(slm.var_), float(slm.regularizer_), float(slm.hypers_))
    slm.obj_, slm.weights_ = fit_obj, fit_m
    out_at = dict(c1_at_elbo=np.array(at["elbo"]), c1_at_dvar=np.array(at["dvar"]), c1_at_dreg=at["dreg"],
                  c1_at_dhyp=np.concatenate(at["dhyp"]), c1_at_m=at["m"], c1_at_Cdiag=at["C"].diagonal().copy())
    out = dict(c1_W_head=b.W[:, :8].copy(), c1_var_=np.array(slm.var_), c1_reg_=np.array(slm.regularizer_),
               c1_hyp_=np.array(slm.hypers_), c1_m=slm.weights_, c1_obj=np.array(slm.obj_), c1_Ey=Ey, c1_Vy=Vy,
               c1_Cdiag=slm.covariance_.diagonal().copy(), c1_ys_true=np.sin(Xs @ np.array([1.0, -0.7, 0.5, 0.3, -0.2, 0.9, -0.4, 0.1])))
    out.update(out_at)
    # second seed of the small end-to-end case, ARD
    N, d, n = 500, 4, 20
    r = np.random.RandomState(17)
    X2 = r.randn(N, d)
    y2 = np.cos(X2 @ np.array([0.8, -0.3, 0.6, 0.2])) + 0.05 * r.randn(N)
    Xs2 = np.random.RandomState(18).randn(16, d)
    b2 = rb.RandomMatern32(nbases=n, Xdim=d, random_state=43, lenscale=Parameter(np.full(d, 1.3), Positive()),
                           regularizer=Parameter(2.0, Positive()))
    s2 = StandardLinearModel(b2, var=Parameter(0.4, Positive()), nstarts=0, maxiter=25)
    s2.fit(X2, y2)
    Ey2, Vy2 = s2.predict_moments(Xs2)
    out.update(s2_X=X2, s2_y=y2, s2_Xs=Xs2, s2_W=b2.W, s2_var_=np.array(s2.var_), s2_reg_=np.array(s2.regularizer_),
               s2_hyp_=np.array(s2.hypers_), s2_obj=np.array(s2.obj_), s2_Ey=Ey2, s2_Vy=Vy2)
    save("fit_c1", **out)


def gen_fit_converged():
    """A CONVERGED reference fit at BASELINE config 1's shape (RandomRBF nbases=256, D=8, N=10k; slm.py:74-140), for the
    estimator tests to be held to at optimiser level.  ARD length scales: with the isotropic one the reference's own gradient
    is the dimension-0 slab only (basis_functions.py:866-901), its L-BFGS-B line search then fails after one or
    two iterations from every start tried ("ABNORMAL"), and the end point is wherever it stalled.  From (var, reg, l) =
    (0.1, 2.0, 4.0 x 8) the reference converges in 26 iterations (and from (0.05, 5.0, 3.0 x 8) to the same optimum: objective
    to 1e-9, hyper-parameters to 1e-4 -- the flat directions are the length scales of inputs that hardly matter).
    `res.success` is asserted through a process-local spy on the `minimize` name revrand.slm imported (nothing is written
    into /root/reference)."""
    import revrand.slm as rslm
    X, y, Xs = c1_data()
    seen = []
    orig = rslm.minimize

    def spy(*a, **k):
        r = orig(*a, **k)
        seen.append(r)
        return r
    rslm.minimize = spy
    try:
        b = rb.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(np.full(8, 4.0), Positive()),
                         regularizer=Parameter(2.0, Positive()))
        slm = StandardLinearModel(b, var=Parameter(0.1, Positive()), nstarts=0, maxiter=500)
        slm.fit(X, y)
    finally:
        rslm.minimize = orig
    res = seen[-1]
    assert res.success and res.nit < 500, res.message
    Ey, Vy = slm.predict_moments(Xs)
    Phi = orc.rff_transform(Xs, b.W, np.asarray(slm.hypers_))
    Eo, Vo = orc.slm_predict_moments(Phi, slm.weights_, slm.covariance_, float(slm.var_))
    close(Ey, Eo, 1e-10)
    close(Vy, Vo, 1e-10)
    ys_true = np.sin(Xs @ np.array([1.0, -0.7, 0.5, 0.3, -0.2, 0.9, -0.4, 0.1]))
    save("fit_converged", W_head=b.W[:, :8].copy(), var_=np.array(slm.var_), reg_=np.array(slm.regularizer_),
         hyp_=np.asarray(slm.hypers_), obj=np.array(slm.obj_), m=slm.weights_, Cdiag=slm.covariance_.diagonal().copy(),
         Ey=Ey, Vy=Vy, ys_true=ys_true, nit=np.array(res.nit), nfev=np.array(res.nfev),
         smse=np.array(((Ey - ys_true) ** 2).mean() / ys_true.var()))


def gen_glm():
    """One minibatch `_elbo` of the reference's GeneralizedLinearModel (glm.py:205-322) per likelihood, with a
    seeded ``random_``; the standard-normal draws are replayed from the same seed and stored so that the oracle
    and the HIP path see the reference's exact samples.  Also a few steps of every SGD updater."""
    from revrand.glm import GeneralizedLinearModel
    import revrand.likelihoods as rl
    import importlib
    rsgd = importlib.import_module("revrand.optimize.sgd")
    out = {}
    rs = np.random.RandomState(11)
    M, d, n, K, Ls = 256, 4, 32, 3, 8
    X = rs.randn(M, d)
    fl = np.sin(X @ rs.randn(d))
    nbin = rs.randint(1, 9, size=M).astype(float)
    ys = {
        "poisson_exp": rs.poisson(np.exp(fl)).astype(float),
        "poisson_softplus": rs.poisson(np.log1p(np.exp(fl))).astype(float),
        "gaussian": fl + 0.1 * rs.randn(M),
        "bernoulli": (rs.rand(M) < 1 / (1 + np.exp(-fl))).astype(float),
        "binomial": rs.binomial(nbin.astype(int), 1 / (1 + np.exp(-fl))).astype(float),
    }
    liks = {
        "poisson_exp": (lambda: rl.Poisson("exp"), [], ()),
        "poisson_softplus": (lambda: rl.Poisson("softplus"), [], ()),
        "gaussian": (lambda: rl.Gaussian(), [0.7], ()),
        "bernoulli": (lambda: rl.Bernoulli(), [], ()),
        "binomial": (lambda: rl.Binomial(), [], (nbin,)),
    }
    D = 2 * n
    m = 0.3 * rs.randn(D, K)
    C = rs.gamma(2., 0.5, size=(D, K))
    out.update(X=X, nbin=nbin, m=m, C=C, K=K, L=Ls, B=10.0, reg=1.3, seed=5)
    for ard in (False, True):
        tag0 = "ard" if ard else "iso"
        lsp = Parameter(np.ones(d), Positive()) if ard else Parameter(1., Positive())
        ls = np.linspace(0.7, 1.4, d) if ard else 0.9
        out[tag0 + "_ls"] = ls
        for name, (mk, lpars, largs) in liks.items():
            if ard and name not in ("poisson_exp", "gaussian"):
                continue
            basis = rb.RandomRBF(nbases=n, Xdim=d, random_state=7, lenscale=lsp)
            glm = GeneralizedLinearModel(likelihood=mk(), basis=basis, K=K, nsamples=Ls, random_state=5)
            glm.B_, glm.D_ = 10.0, D
            glm._GeneralizedLinearModel__it = -1
            y = ys[name]
            lp = lpars[0] if lpars else []
            nobj, (ndm, ndC, dL, dlp, dbp) = glm._elbo(m.copy(), C.copy(), 1.3, lp, ls, X, y, *largs)
            e = np.stack([np.random.RandomState(5).randn(K * Ls, D)[k * Ls:(k + 1) * Ls] for k in range(K)])
            Phi = basis.transform(X, ls)
            dP = basis.grad(X, ls)
            dPs = [dP[:, :, i] for i in range(d)] if ard else [dP]
            o = orc.glm_elbo(m, C, np.full(D, 1.3), slice(None), name, lpars, largs, Phi, dPs, y, e, 10.0)
            close(o[0], nobj, 1e-10)
            close(o[1][0], ndm, 1e-10)
            close(o[1][1], ndC, 1e-10)
            close(o[1][2][0], dL, 1e-10)
            if lpars:
                close(o[1][3][0], np.atleast_1d(dlp)[0], 1e-10)
            close(np.array(o[1][4]), np.atleast_1d(dbp), 1e-10)
            t = tag0 + "_" + name
            out.update({t + "_y": y, t + "_obj": nobj, t + "_ndm": ndm, t + "_ndC": ndC, t + "_dL": dL,
                        t + "_dlp": np.atleast_1d(np.asarray(dlp, float)) if lpars else np.zeros(0),
                        t + "_dbp": np.atleast_1d(dbp)})
            out["W"] = basis.W
    out["e"] = np.stack([np.random.RandomState(5).randn(K * Ls, D)[k * Ls:(k + 1) * Ls] for k in range(K)])
    # SGD updaters: five steps on a fixed gradient sequence
    gs = np.random.RandomState(3).randn(5, 6)
    for name, upd in (("sgd", rsgd.SGDUpdater()), ("adadelta", rsgd.AdaDelta()), ("adagrad", rsgd.AdaGrad()),
                      ("momentum", rsgd.Momentum()), ("adam", rsgd.Adam())):
        x = np.linspace(-1, 1, 6)
        xo, st = x.copy(), {}
        traj = []
        for g in gs:
            x = upd(x, g)
            xo = orc.sgd_update(name, st, xo, g)
            traj.append(x.copy())
        close(xo, x, 1e-12)
        out["upd_" + name] = np.array(traj)
    out["upd_grads"] = gs
    save("glm", **out)


def gen_glm_fit():
    """End-to-end `GeneralizedLinearModel.fit` of the reference (glm.py:141-203): random starts (decorators.py:541-583),
    structured_sgd / logtrick_sgd (decorators.py:133-252, 329-408), sgd with bounds (optimize/sgd.py:337-425) and the
    interleaving of `gen_batch` permutations with `_reparam_k` draws on ONE RandomState -- 20 Adam steps on small data,
    seeds fixed (the estimator's `random_state` and NumPy's global stream, from which the reference draws the start
    point).  Stored per case: data, W of every Fourier child, the five fitted blocks, the optimiser's objective / norm
    records and `random_.randn()` after the fit (the stream's end state).

    `bs64f` cases: the reference's structured_sgd does not forward `batch_size` to sgd (decorators.py:244-246), so its
    main loop always runs sgd's default of 10 rows while B_ = N / batch_size.  The implementation here forwards it.  For
    those cases the REFERENCE'S OWN code is run with sgd's default batch size set to the estimator's (`functools.partial`
    on the name `revrand.glm` imported, process-local); `bs64` (no `f`) is the unmodified reference at batch_size=64."""
    import functools
    from scipy.stats import gamma
    import revrand.glm as rglm
    import revrand.likelihoods as rl
    from revrand.btypes import Bound
    real_sgd = rglm.sgd
    out = {}
    rs = np.random.RandomState(21)
    N, d, n = 300, 3, 8
    X = rs.randn(N, d)
    f = 0.6 * np.sin(X[:, 0]) + 0.3 * X[:, 1]
    nbin = rs.randint(4, 12, size=N).astype(float)
    ys = {"poisson_exp": rs.poisson(np.exp(f)).astype(float), "gaussian": f + 0.1 * rs.randn(N),
          "binomial": rs.binomial(nbin.astype(int), 1 / (1 + np.exp(-2 * f))).astype(float)}
    ys["bernoulli"] = (np.random.RandomState(22).rand(N) < 1 / (1 + np.exp(-3 * f))).astype(float)   # (own stream: the others' data stay as they were)
    ys["poisson_softplus"] = ys["poisson_exp"]
    out.update(X=X, nbin=nbin, **{"y_" + k: v for k, v in ys.items() if k != "poisson_softplus"})
    K, L, maxiter, seed, gseed = 3, 6, 20, 17, 5
    out.update(K=K, L=L, maxiter=maxiter, seed=seed, global_seed=gseed, nbases=n)

    def ard_rbf():
        return rb.RandomRBF(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(gamma(4., scale=0.25), Positive(), shape=(d,)))

    def concat():  # tests/test_models.py:97-99
        return rb.LinearBasis(onescol=True) + rb.RandomRBF(nbases=n, Xdim=d, random_state=3) \
            + rb.RandomMatern52(nbases=n, Xdim=d, random_state=4)

    def bounded():  # a plain Bound (no log trick) that 20 Adam steps of 0.01 run into
        return rb.RandomRBF(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(1.0, Bound(0.996, 1.001)))

    def posupper():  # Positive with an upper limit: the log trick's upper bound log(upper)
        return rb.RandomRBF(nbases=n, Xdim=d, random_state=3, lenscale=Parameter(1.0, Positive(1.03)))

    def child_specs(basis):
        kids = basis.bases if hasattr(basis, "bases") else [basis]
        ch, regs, lss = [], [], []
        for b in kids:
            r = b.regularizer
            regs.append(orc.ParamSpec(dist=r.dist, value=None if r.dist is not None else r.value, positive=True, shape=r.shape))
            if isinstance(b, rb.LinearBasis):
                ch.append(("linear", b.onescol))
                lss.append(orc.ParamSpec(value=[]))
            else:
                p = b.params
                ch.append(("rff", b.W, int(np.prod(p.shape, dtype=int))))
                lss.append(orc.ParamSpec(dist=p.dist, value=None if p.dist is not None else p.value,
                                         positive=isinstance(p.bounds, Positive), lower=None if isinstance(p.bounds, Positive) else p.bounds.lower,
                                         upper=p.bounds.upper, shape=p.shape))
        return ch, regs, lss

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from glm_fit_cases import CASES as cases, updater_of
    import importlib
    rsgd = importlib.import_module("revrand.optimize.sgd")
    bas = {"ard": ard_rbf, "cat": concat, "bound": bounded, "posupper": posupper}
    cases = [(t, l, bas[b], bs_, ns_, f) for t, l, b, bs_, ns_, f in cases]
    mk = {"poisson_exp": lambda: rl.Poisson("exp"), "gaussian": rl.Gaussian, "binomial": rl.Binomial, "bernoulli": rl.Bernoulli,
          "poisson_softplus": lambda: rl.Poisson("softplus")}
    for tag, lik, mkbasis, bs, ns, fwd in cases:
        basis = mkbasis()
        like = mk[lik]()
        largs = (nbin,) if lik == "binomial" else ()
        cap = []

        def spy(*a, **k):
            if fwd:
                k.setdefault("batch_size", bs)
            res = real_sgd(*a, **k)
            cap.append(res)
            return res
        rglm.sgd = spy
        try:
            upd = updater_of(tag)
            glm = rglm.GeneralizedLinearModel(like, basis, K=K, nsamples=L, batch_size=bs, maxiter=maxiter, nstarts=ns,
                                              random_state=seed, updater=None if upd is None else getattr(rsgd, upd[0])(**upd[1]))
            np.random.seed(gseed)
            glm.fit(X, ys[lik], likelihood_args=largs)
        finally:
            rglm.sgd = real_sgd
        end = glm.random_.randn()
        res = cap[0]
        ch, regs, lss = child_specs(basis)
        lp = like.params
        likpar = [orc.ParamSpec(dist=lp.dist, positive=True, shape=lp.shape)] if lik == "gaussian" else []
        o = orc.glm_fit(X, ys[lik], lik, list(largs), ch, regs, likpar, lss, K, L, bs, maxiter, ns, seed, gseed,
                        sgd_batch_size=bs if fwd else 10, **({} if upd is None else {"updater": upd[2], "updater_hp": upd[1]}))
        flat = lambda v: np.concatenate([np.ravel(np.asarray(u, float)) for u in (v if isinstance(v, (list, tuple)) else [v])] + [np.empty(0)])
        close(o[0], glm.weights_, 1e-9)
        close(o[1], glm.covariance_, 1e-9)
        close(flat(o[2]), flat(glm.regularizer_), 1e-9)
        close(flat(o[3]), flat(glm.like_hypers_), 1e-9)
        close(flat(o[4]), flat(glm.basis_hypers_), 1e-9)
        close(o[6], np.array(res.norms), 1e-9)
        fin = np.isfinite(np.array(res.objs, float))
        assert np.array_equal(fin, np.isfinite(o[5])) and fin.any()
        close(o[5][fin], np.array(res.objs, float)[fin], 1e-9)
        assert o[7] == end, (o[7], end)
        out.update({tag + "_m": glm.weights_, tag + "_C": glm.covariance_, tag + "_reg": flat(glm.regularizer_),
                    tag + "_lik": flat(glm.like_hypers_), tag + "_ls": flat(glm.basis_hypers_),
                    tag + "_norms": np.array(res.norms), tag + "_objs": np.array(res.objs, float), tag + "_end": np.array(end)})
        for i, c in enumerate(ch):
            if c[0] == "rff":
                out["%s_W%d" % (tag, i)] = c[1]
        print("   ", tag, "||g|| first/last %.4g %.4g" % (res.norms[0], res.norms[-1]),
              "ls", np.round(flat(glm.basis_hypers_), 4))
    save("glm_fit", **out)


def gen_glm_predict():
    """The prediction surface of the reference's GeneralizedLinearModel on a model whose fitted attributes are SET (no fit):
    `_sample_func` (glm.py:572-620: k ~ randint(K), w = m_k + randn sqrt(C_k), f = Phi w), `predict_moments` (:351-393),
    `predict_logpdf` (:395-444), `predict_cdf` (:446-495) and `predict_interval` (:497-570, brentq per row) for a Gaussian, a
    Poisson and a binomial likelihood, `random_` re-seeded before every call."""
    import revrand.glm as rglm
    import revrand.likelihoods as rl
    out = {}
    rs = np.random.RandomState(31)
    N, d, n, K, S = 40, 3, 6, 3, 50
    X = rs.randn(N, d)
    nbin = rs.randint(5, 15, size=N).astype(float)
    basis = rb.LinearBasis(onescol=True) + rb.RandomRBF(nbases=n, Xdim=d, random_state=8)
    D = d + 1 + 2 * n
    m = 0.4 * rs.randn(D, K)
    C = rs.gamma(2., 0.05, size=(D, K))
    ls = 1.3
    yq = {"gaussian": rs.randn(N), "poisson_exp": rs.poisson(2.0, N).astype(float), "binomial": rs.binomial(nbin.astype(int), 0.4).astype(float)}
    out.update(X=X, nbin=nbin, m=m, C=C, ls=ls, K=K, S=S, W=basis.bases[1].W, **{"yq_" + k: v for k, v in yq.items()})
    mk = {"gaussian": (rl.Gaussian, [0.3]), "poisson_exp": (lambda: rl.Poisson("exp"), []), "binomial": (rl.Binomial, [])}
    for lik, (ctor, lhyp) in mk.items():
        glm = rglm.GeneralizedLinearModel(ctor(), basis, K=K, random_state=0)
        glm.weights_, glm.covariance_, glm.regularizer_ = m, C, [1.0, 1.0]
        glm.like_hypers_ = lhyp[0] if lhyp else []
        glm.basis_hypers_ = ls
        largs = (nbin,) if lik == "binomial" else ()

        def seeded(fn, *a, **k):
            glm.random_ = np.random.RandomState(77)
            return fn(*a, **k)
        fs = np.array(list(seeded(glm._sample_func, X, S)))                       # (S, N)
        Ey, Vy = seeded(glm.predict_moments, X, S, likelihood_args=largs)
        lp = seeded(glm.predict_logpdf, X, yq[lik], S, likelihood_args=largs)
        q = 2.5 if lik != "gaussian" else 0.2
        cdf = seeded(glm.predict_cdf, X, q, S, likelihood_args=largs)
        ql, qu = seeded(glm.predict_interval, X[:12], 0.9, S, likelihood_args=tuple(a[:12] for a in largs), multiproc=False)
        # the oracle's restatement on the same draws
        o = orc.glm_predictions(X, m, C, [("linear", True), ("rff", basis.bases[1].W, 1)], [[], ls], lik, lhyp, list(largs), S, 77,
                                yq[lik], q)
        close(o["fs"], fs, 1e-12)
        close(o["Ey"], Ey, 1e-12); close(o["Vy"], Vy, 1e-12)
        for a, b in zip(o["logpdf"], lp):
            close(a, b, 1e-12)
        for a, b in zip(o["cdf"], cdf):
            close(a, b, 1e-12)
        out.update({lik + "_fs": fs, lik + "_Ey": Ey, lik + "_Vy": Vy, lik + "_logpdf": np.array(lp), lik + "_cdf": np.array(cdf),
                    lik + "_q": q, lik + "_ql": ql, lik + "_qu": qu})
        print("   ", lik, "Ey[:3]", np.round(Ey[:3], 4), "interval[0]", ql[0], qu[0])
    save("glm_predict", **out)


def gen_slm_starts():
    """The random starts of `StandardLinearModel.fit` (slm.py:74-140 -> structured_minimizer, decorators.py:24-130, 541-583): with
    distributions as initial values the reference first draws a start from `random_` (the flatten of :79-84), then `nstarts`
    candidates, each scored by `_elbo`, and hands the best to L-BFGS-B.  maxiter = 0 makes `fit` return that start: stored
    with every candidate's objective (spy on `_elbo`) and `random_.randn()` afterwards."""
    from scipy.stats import gamma
    out = {}
    rs = np.random.RandomState(41)
    N, d, n = 300, 3, 10
    X = rs.randn(N, d)
    y = np.sin(X @ np.array([1.0, -0.5, 0.3])) + 0.1 * rs.randn(N)
    out.update(X=X, y=y)
    for tag, ns in (("ns6", 6), ("ns1", 1)):
        basis = rb.RandomRBF(nbases=n, Xdim=d, random_state=9, lenscale=Parameter(gamma(2., scale=0.5), Positive(), shape=(d,)),
                             regularizer=Parameter(gamma(1.), Positive()))
        slm = StandardLinearModel(basis, var=Parameter(gamma(1.), Positive()), nstarts=ns, maxiter=0, random_state=13)
        objs = []
        real = StandardLinearModel._elbo

        def spy(self, *a, **k):
            r = real(self, *a, **k)
            objs.append(r[0])
            return r
        StandardLinearModel._elbo = spy
        try:
            slm.fit(X, y)
        finally:
            StandardLinearModel._elbo = real
        assert len(objs) >= ns + 1
        out.update({tag + "_var": slm.var_, tag + "_reg": slm.regularizer_, tag + "_hyp": np.asarray(slm.hypers_),
                    tag + "_cand_objs": np.array(objs[:ns]), tag + "_end": slm.random_.randn(), "W": basis.W})
        print("   ", tag, "start var %.4f reg %.4f hyp %s; best of" % (slm.var_, slm.regularizer_, np.round(slm.hypers_, 3)), np.round(objs[:ns], 2))
    save("slm_starts", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:  # regenerate selected fixtures only, e.g. `make_golden.py fit_c1`
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_weights()
    gen_rff()
    gen_hadamard()
    gen_fastfood()
    gen_fastfood_gm()
    gen_concat()
    gen_elbo()
    gen_solve_posdef()
    gen_fit()
    gen_fit_c1()
    gen_fit_converged()
    gen_glm()
    gen_glm_fit()
    gen_glm_predict()
    gen_slm_starts()
    print("oracle agrees with the reference on every fixture")
