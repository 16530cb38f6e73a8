"""
CPU oracle for the revrand random-feature hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement (written from the maths in SURVEY.md section 8a,
not copied) of what NICTA/revrand computes on the path

    basis.transform / basis.grad  ->  Phi^T Phi, Phi^T y  ->  posterior solve / ELBO

It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``revrand_amd/`` (the product) imports, links or executes anything in this
directory; the product path runs on the HIP library and fails loudly without it.

Pinning: every function here is checked against outputs of the reference itself
(imported from /root/reference in the build container by ``oracle/make_golden.py``)
which are committed as fixtures under ``tests/golden/``; ``hadamard`` is also
pinned by the reference's own doctest vector (revrand/mathfun/linalg.py:202-206).
See ``tests/test_oracle_golden.py``.

All citations ``file:line`` are into the reference tree (revrand v1.0.0).
"""

import numpy as np
from scipy import linalg as sla

CHOLTHRESH = 1e-5  # revrand/mathfun/linalg.py:31


# --------------------------------------------------------------------------
# a-3  frequency matrices (host-side sampling, MT19937 legacy stream)
# --------------------------------------------------------------------------

def _rs(seed_or_state):
    if isinstance(seed_or_state, np.random.RandomState):
        return seed_or_state
    return np.random.RandomState(seed_or_state)


def weights_rbf(d, n, seed):
    """RandomRBF._weightsamples  basis_functions.py:952-954 : randn(d, n)."""
    return _rs(seed).randn(d, n)


def weights_laplace(d, n, seed):
    """RandomLaplace._weightsamples  basis_functions.py:993-995 : Cauchy draws."""
    return _rs(seed).standard_cauchy(size=(d, n))


def weights_cauchy(d, n, seed):
    """RandomCauchy._weightsamples  basis_functions.py:1034-1045.

    Gaussian draw scaled per column by sqrt(2 * Gamma(1)) (a Laplace-mixture).
    """
    r = _rs(seed)
    g = r.randn(d, n)
    z = r.standard_gamma(1., size=(1, n))
    return g * np.sqrt(2. * z)


def weights_matern(d, n, seed, p):
    """_RandomMatern._maternweight  basis_functions.py:1051-1065.

    Multivariate-t draw with df = 2p+1: normal / sqrt(chi2/df), chi2 per column.
    p=1 -> Matern32 (:1107), p=2 -> Matern52 (:1150).
    """
    r = _rs(seed)
    df = 2. * (p + 0.5)
    g = r.randn(d, n)
    u = r.chisquare(df, size=(n,))
    return g * np.sqrt(df / u)


def weights_orthogonal(d, n, seed):
    """OrthogonalRBF._weightsamples  basis_functions.py:1198-1208.

    ceil(n/d) QR blocks of a (d,d) Gaussian, columns truncated to n, rows scaled
    by sqrt(chi2_d) draws (drawn AFTER all the Gaussian blocks).
    """
    r = _rs(seed)
    reps = int(np.ceil(n / d))
    blocks = []
    for _ in range(reps):
        q, _r = sla.qr(r.randn(d, d))
        blocks.append(q)
    Q = np.concatenate(blocks, axis=1)[:, :n]
    s = np.sqrt(r.chisquare(df=d, size=d))
    return s[:, None] * Q


# --------------------------------------------------------------------------
# a-1 / a-2  random Fourier features
# --------------------------------------------------------------------------

def _lenscale_col(lenscale, d):
    ls = np.atleast_1d(np.asarray(lenscale, dtype=float))
    if ls.shape not in ((1,), (d,)):
        raise ValueError("Dimension of input parameter is inconsistent!")
    return ls


def rff_transform(X, W, lenscale):
    """_RandomKernelBasis.transform  basis_functions.py:838-864.

    Phi = [cos(X (W/l)), sin(X (W/l))] / sqrt(n), cos block first, always float64.
    """
    X = np.asarray(X)
    d, n = W.shape
    ls = _lenscale_col(lenscale, d)
    Z = X @ (W / ls[:, None])
    out = np.empty((X.shape[0], 2 * n))
    np.cos(Z, out=out[:, :n])
    np.sin(Z, out=out[:, n:])
    out /= np.sqrt(n)
    return out


def rff_grad(X, W, lenscale):
    """_RandomKernelBasis.grad  basis_functions.py:866-901.

    For each entry i of the *given* lenscale vector (ONE entry when isotropic --
    the reference then only differentiates through input dimension 0, a quirk
    parity must keep):
        dZ_i = -outer(X[:, i], W[i, :]) / l_i^2
        dPhi_i = [ -sin(Z) * dZ_i , cos(Z) * dZ_i ] / sqrt(n)
    Returns (N, 2n) for one lenscale, (N, 2n, d) for ARD.
    """
    X = np.asarray(X)
    d, n = W.shape
    ls = _lenscale_col(lenscale, d)
    Z = X @ (W / ls[:, None])
    ms, c = -np.sin(Z), np.cos(Z)
    scale = 1. / np.sqrt(n)
    slabs = []
    for i, l in enumerate(ls):
        dZ = -(X[:, i:i + 1] * W[i:i + 1, :]) / l ** 2
        slabs.append(np.concatenate((dZ * ms, dZ * c), axis=1) * scale)
    if len(slabs) == 1:
        return slabs[0]
    return np.stack(slabs, axis=2)


# --------------------------------------------------------------------------
# a-7  Walsh-Hadamard transform
# --------------------------------------------------------------------------

def hadamard(Y, ordering=False):
    """mathfun.linalg.hadamard  linalg.py:182-220.

    Natural(Hadamard)-order WHT normalised by 1/n  ( == Y @ H_n / n ).  Radix-2
    butterflies, each stage halved.  ``ordering=True`` applies the sequency
    permutation of linalg.py:223-236.
    """
    Y = np.array(Y, dtype=float)
    rows, n = Y.shape
    if n & (n - 1):
        raise AssertionError("length must be a power of two")
    h = 1
    while h < n:
        V = Y.reshape(rows, n // (2 * h), 2, h)
        a = V[:, :, 0, :].copy()
        b = V[:, :, 1, :].copy()
        V[:, :, 0, :] = (a + b) * 0.5
        V[:, :, 1, :] = (a - b) * 0.5
        h *= 2
    if ordering:
        Y = Y[:, sequency(n)]
    return Y


def sequency(n):
    """mathfun.linalg._sequency  linalg.py:223-236: bit-reversed Gray code."""
    bits = int(np.log2(n))
    idx = np.arange(n)
    gray = idx ^ (idx >> 1)
    out = np.zeros(n, dtype=int)
    for b in range(bits):
        out |= ((gray >> b) & 1) << (bits - 1 - b)
    return out


# --------------------------------------------------------------------------
# a-5 / a-6  FastFood
# --------------------------------------------------------------------------

def fastfood_dims(nbases, d):
    """FastFoodRBF._init_dims  basis_functions.py:1331-1340 -> (d2, k, n)."""
    d2 = 1 << int(np.ceil(np.log2(d)))
    k = int(np.ceil(nbases / d2))
    return d2, k, d2 * k


def fastfood_matrices(nbases, d, seed):
    """FastFoodRBF._init_matrices/_weightsamples  basis_functions.py:1342-1354.

    Draw order B -> G -> PI -> S.  S = d2 * sqrt(chi2_{d2}) / ||G_row||_2.
    """
    r = _rs(seed)
    d2, k, _n = fastfood_dims(nbases, d)
    B = r.randint(2, size=(k, d2)) * 2 - 1
    G = r.randn(k, d2)
    PI = np.array([r.permutation(d2) for _ in range(k)])
    chi = np.sqrt(r.chisquare(d2, size=(k, d2)))
    S = d2 * chi / np.sqrt((G ** 2).sum(axis=1))[:, None]
    return B, G, PI, S


def fastfood_VX(X, B, G, PI, S):
    """FastFoodRBF._makeVX  basis_functions.py:1356-1371.

    Per block: v = H(x~ * B); v = v[PI] * G; v = H(v) * S * sqrt(d2), H = WHT/d2.
    """
    N, d0 = X.shape
    k, d2 = B.shape
    Xp = np.zeros((N, d2))
    Xp[:, :d0] = X
    out = np.empty((N, k * d2))
    root = np.sqrt(d2)
    for j in range(k):
        v = hadamard(Xp * B[j], ordering=False)
        v = v[:, PI[j]] * G[j]
        out[:, j * d2:(j + 1) * d2] = hadamard(v, ordering=False) * S[j] * root
    return out


def fastfood_transform(X, B, G, PI, S, lenscale):
    """FastFoodRBF.transform  basis_functions.py:1263-1289."""
    d = X.shape[1]
    ls = _lenscale_col(lenscale, d)
    V = fastfood_VX(X / ls, B, G, PI, S)
    n = V.shape[1]
    return np.concatenate((np.cos(V), np.sin(V)), axis=1) / np.sqrt(n)


def fastfood_grad(X, B, G, PI, S, lenscale):
    """FastFoodRBF.grad  basis_functions.py:1291-1329 (same iso quirk as RFF)."""
    d = X.shape[1]
    ls = _lenscale_col(lenscale, d)
    V = fastfood_VX(X / ls, B, G, PI, S)
    n = V.shape[1]
    ms, c = -np.sin(V), np.cos(V)
    slabs = []
    for i, l in enumerate(ls):
        e = np.zeros(d)
        e[i] = 1. / l ** 2
        dV = -fastfood_VX(X * e, B, G, PI, S)
        slabs.append(np.concatenate((dV * ms, dV * c), axis=1) / np.sqrt(n))
    if len(slabs) == 1:
        return slabs[0]
    return np.stack(slabs, axis=2)


def fastfood_gm_transform(X, B, G, PI, S, mean, lenscale):
    """FastFoodGM.transform  basis_functions.py:1443-1475: four trig blocks of VX +- X.mean, / sqrt(2n)."""
    d = X.shape[1]
    ls = _lenscale_col(lenscale, d)
    mu = _lenscale_col(mean, d)
    V = fastfood_VX(X / ls, B, G, PI, S)
    mX = (X @ mu)[:, None]
    n = V.shape[1]
    return np.concatenate((np.cos(V + mX), np.sin(V + mX), np.cos(V - mX), np.sin(V - mX)), axis=1) / np.sqrt(2 * n)


def fastfood_gm_grad(X, B, G, PI, S, mean, lenscale):
    """FastFoodGM.grad  basis_functions.py:1477-1537 -> (dPhi/dmean, dPhi/dlenscale), each (N, 4n, d)
    ((N, 4n) when d == 1)."""
    d = X.shape[1]
    ls = _lenscale_col(lenscale, d)
    mu = _lenscale_col(mean, d)
    V = fastfood_VX(X / ls, B, G, PI, S)
    mX = (X @ mu)[:, None]
    n = V.shape[1]
    msp, msm, cp, cm = -np.sin(V + mX), -np.sin(V - mX), np.cos(V + mX), np.cos(V - mX)
    rt = np.sqrt(2 * n)
    gm, gl = [], []
    for i, l in enumerate(ls):
        xi = X[:, i:i + 1]
        gm.append(np.concatenate((xi * msp, xi * cp, -xi * msm, -xi * cm), axis=1) / rt)
        e = np.zeros(d)
        e[i] = 1. / l ** 2
        dV = -fastfood_VX(X * e, B, G, PI, S)
        gl.append(np.concatenate((dV * msp, dV * cp, dV * msm, dV * cm), axis=1) / rt)
    if d == 1:
        return gm[0], gl[0]
    return np.stack(gm, axis=2), np.stack(gl, axis=2)


# --------------------------------------------------------------------------
# a-8  LinearBasis / concatenation
# --------------------------------------------------------------------------

def linear_transform(X, onescol=True):
    """LinearBasis.transform  basis_functions.py:468-485."""
    X = np.asarray(X, dtype=float)
    if not onescol:
        return X
    return np.concatenate((np.ones((X.shape[0], 1)), X), axis=1)


# --------------------------------------------------------------------------
# a-11  Gram statistics  (slm.py:145-146,157)
# --------------------------------------------------------------------------

def gram_stats(Phi, y):
    """G = Phi^T Phi, b = Phi^T y, yty = y^T y   (slm.py:146,157,161-162)."""
    return Phi.T @ Phi, Phi.T @ y, float(y @ y)


def rff_gram_chunked(X, y, W, lenscale, chunk=10000, dtype=np.float64):
    """Chunk-accumulated Phi + Phi^T Phi + Phi^T y: the metric's unit of work
    exactly as revrand's NumPy path executes it (basis_functions.py:859-864,
    slm.py:145-146,157), restated so that Phi for large N need not be held.
    This is what bench.py times as ``cpu_baseline`` (kind "port")."""
    d, n = W.shape
    ls = _lenscale_col(lenscale, d)
    Ws = (W / ls[:, None]).astype(dtype)
    F = 2 * n
    Gm = np.zeros((F, F))
    b = np.zeros(F)
    yty = 0.0
    rn = 1. / np.sqrt(n)
    for s in range(0, X.shape[0], chunk):
        Xc = X[s:s + chunk]
        yc = y[s:s + chunk]
        Z = np.dot(Xc, Ws)
        P = np.hstack((np.cos(Z), np.sin(Z))) * rn
        Gm += P.T.dot(P)
        b += P.T.dot(yc)
        yty += float(yc.dot(yc))
    return Gm, b, yty


# --------------------------------------------------------------------------
# a-13  solve_posdef
# --------------------------------------------------------------------------

def solve_posdef(A, B):
    """mathfun.linalg.solve_posdef  linalg.py:84-125 (+ svd_solve :128-179).

    Upper Cholesky; if it fails or any diagonal < CHOLTHRESH use the SVD with
    singular values clamped at 1e-15.  Returns (A^-1 B, log|A|).
    """
    try:
        U = sla.cholesky(A, lower=False)
        if np.any(U.diagonal() < CHOLTHRESH):
            raise sla.LinAlgError("unstable")
        X = sla.cho_solve((U, False), B)
        logdet = 2. * np.log(U.diagonal()).sum()
    except sla.LinAlgError:
        Us, s, Vt = sla.svd(A)
        isq = 1. / np.sqrt(np.maximum(s, 1e-15))
        X = (Us * isq) @ ((isq[:, None] * Vt) @ B)
        logdet = np.log(s).sum()
    return X, logdet


# --------------------------------------------------------------------------
# a-11 / a-12  ELBO of the standard linear model
# --------------------------------------------------------------------------

def slm_elbo(Phi, y, var, reg_diag, slices, dPhis):
    """StandardLinearModel._elbo  slm.py:142-199, on a materialised Phi.

    reg_diag : (F,) prior variance diagonal (basis.regularizer_diagonal)
    slices   : slice or list of slices (one per concatenated basis)
    dPhis    : list of 2-D (N,F) gradient slabs (already split along the ARD axis)

    Returns dict(elbo, m, C, logdetC, dvar, dreg(list), dhyp(list)).  All
    gradients are of +ELBO; the reference hands the optimiser the gradients of
    -ELBO (``-dvar`` at slm.py:199, and ``dreg``/``dhyps`` are already defined
    with the minus sign folded in, slm.py:187,194).
    """
    N, F = Phi.shape
    G = Phi.T @ Phi
    iL = 1. / reg_diag
    iC = np.diag(iL) + G / var
    C, logdet_iC = solve_posdef(iC, np.eye(F))
    logdetC = -logdet_iC
    m = C @ (Phi.T @ y) / var
    trGC = (G * C).sum()
    err = y - Phi @ m
    sq = (err ** 2).sum()
    elbo = -0.5 * (N * np.log(2 * np.pi * var) + sq / var + trGC / var
                   + ((m ** 2 + C.diagonal()) * iL).sum()
                   - logdetC + np.log(reg_diag).sum() - F)
    dvar = 0.5 * (-N + (sq + trGC) / var) / var
    sl = slices if isinstance(slices, (list, tuple)) else [slices]
    dreg = [0.5 * (((m[s] ** 2 + C[s, s].diagonal()) * iL[s] ** 2).sum()
                   - iL[s].sum()) for s in sl]
    # slm.py:193-195 has (dPhi.T.dot(Phi) * C).sum() per slab -- an (F, F) product per length scale: 1 s per slab at
    # F = 4096, 28 s of a 32-slab check.  The same number by the trace identity
    #     sum((dPhi^T Phi) o C) = sum_r dPhi_r . (C Phi_r) = sum(dPhi o (Phi C))          (C symmetric)
    # with Phi C formed ONCE; small cases keep the reference's own expression (tests/test_oracle_golden.py holds both to the
    # reference's outputs).
    if F <= 512:
        dhyp = [(m @ (err @ dP) - ((dP.T @ Phi) * C).sum()) / var for dP in map(np.ascontiguousarray, dPhis)]
    else:
        PC = Phi @ C
        dhyp = [(m @ (err @ dP) - (dP * PC).sum()) / var for dP in dPhis]
    return dict(elbo=elbo, m=m, C=C, logdetC=logdetC, dvar=dvar, dreg=dreg,
                dhyp=dhyp, G=G, b=Phi.T @ y)


def slm_posterior_from_stats(G, b, var, reg_diag):
    """Posterior (m, C, logdetC) from sufficient statistics (slm.py:154-157)."""
    F = G.shape[0]
    iC = np.diag(1. / reg_diag) + G / var
    C, logdet_iC = solve_posdef(iC, np.eye(F))
    return C @ b / var, C, -logdet_iC


def slm_predict_moments(Phi_s, m, C, var):
    """StandardLinearModel.predict_moments  slm.py:219-244."""
    return Phi_s @ m, ((Phi_s @ C) * Phi_s).sum(axis=1) + var


# --------------------------------------------------------------------------
# a-15  GeneralizedLinearModel._elbo (one SVI minibatch), likelihood derivatives, SGD updaters
# --------------------------------------------------------------------------

def softplus(f):
    """mathfun/special.py:91-128: log(1 + exp(f)) through logsumexp([0, f])."""
    f = np.asarray(f, dtype=float)
    return np.maximum(f, 0.) + np.log1p(np.exp(-np.abs(f)))


def _expit(f):
    f = np.asarray(f, dtype=float)
    return np.where(f >= 0, 1. / (1. + np.exp(-np.abs(f))), np.exp(-np.abs(f)) / (1. + np.exp(-np.abs(f))))


def lik_loglike(name, y, f, *args):
    """likelihoods.py: Bernoulli :46-67, Binomial :171-192, Gaussian :298-323, Poisson :456-481."""
    from scipy.special import gammaln
    y, f = np.broadcast_arrays(np.asarray(y, float), np.asarray(f, float))
    if name == "bernoulli":
        return y * f - softplus(f)
    if name == "binomial":
        n = np.broadcast_to(np.asarray(args[0], float), f.shape)
        return gammaln(n + 1) - gammaln(y + 1) - gammaln(n - y + 1) + y * f - n * softplus(f)
    if name == "gaussian":
        var = args[0]
        return -0.5 * (np.log(2 * np.pi * var) + (y - f) ** 2 / var)
    if name == "poisson_exp":
        return y * f - np.exp(f) - gammaln(y + 1)
    if name == "poisson_softplus":
        g = softplus(f)
        return y * np.log(g) - g - gammaln(y + 1)
    raise ValueError(name)


def lik_df(name, y, f, *args):
    """d loglike / d f: likelihoods.py :86-104, :213-233, :346-368, :500-521."""
    y, f = np.broadcast_arrays(np.asarray(y, float), np.asarray(f, float))
    if name == "bernoulli":
        return y - _expit(f)
    if name == "binomial":
        return y - _expit(f) * np.broadcast_to(np.asarray(args[0], float), f.shape)
    if name == "gaussian":
        return (y - f) / args[0]
    if name == "poisson_exp":
        return y - np.exp(f)
    if name == "poisson_softplus":
        return _expit(f) * (y / np.maximum(softplus(f), 1e-100) - 1)
    raise ValueError(name)


def lik_dp(name, y, f, *args):
    """d loglike / d likelihood-parameter: only the Gaussian has one (likelihoods.py:370-396)."""
    if name != "gaussian":
        return []
    y, f = np.broadcast_arrays(np.asarray(y, float), np.asarray(f, float))
    ivar = 1. / args[0]
    return [0.5 * (((y - f) * ivar) ** 2 - ivar)]


def glm_qmatrix(m, C):
    """glm.py:697-712: logq[j, i] = log N(m_i | m_j, diag(C_i + C_j))."""
    D, K = m.shape
    q = np.empty((K, K))
    for j in range(K):
        for i in range(K):
            dc = C[:, i] + C[:, j]
            q[j, i] = -0.5 * (D * np.log(2 * np.pi) + np.log(dc).sum() + ((m[:, i] - m[:, j]) ** 2 / dc).sum())
    return q


def glm_elbo(m, C, reg_diag, slices, lik, lpars, largs, Phi, dPhis, y, e, B, calc_ll=True):
    """GeneralizedLinearModel._elbo + _reparam_k  glm.py:205-322 for one minibatch, on a materialised Phi,
    with the standard-normal draws e (K, L, D) given (the reference draws ``random_.randn(L, D)`` once per
    mixture component, in order, glm.py:300).

    Returns what the reference hands its optimiser: (-ELBO, [-dm, -dC, dL, dlpars, dbpars]) with dL a list
    (one per slice), dlpars a list, dbpars a list (one per 2-D slab in dPhis).
    """
    D, K = m.shape
    L = e.shape[1]
    pars = list(lpars) + list(largs)
    Ld = np.asarray(reg_diag, float)
    iL = 1. / Ld[:, None]
    logNkl = glm_qmatrix(m, C)
    mx = logNkl.max(axis=0)
    logzk = np.log(np.exp(logNkl - mx).sum(axis=0)) + mx
    dm, dC, Ell = np.empty_like(m), np.empty_like(C), np.empty(K)
    dlpars = [np.zeros_like(np.asarray(p, float)) for p in lpars]
    EdPhi = np.zeros_like(Phi)
    for k in range(K):
        Sk = np.sqrt(C[:, k])
        ws = m[:, k] + Sk * e[k]
        fs = ws @ Phi.T
        dfs = lik_df(lik, y, fs, *pars)
        Edws = dfs @ Phi
        Edm = Edws.sum(axis=0) / L
        EdC = (Edws * e[k] / Sk).sum(axis=0) / L
        EdPhi += (dfs.T @ ws / L) / K
        for i, dp in enumerate(lik_dp(lik, y, fs, *pars)):
            dlpars[i] = dlpars[i] - dp.sum() / L / K
        Ell[k] = lik_loglike(lik, y, fs, *pars).sum() / L if calc_ll else np.inf
        Nkl_zk = np.exp(logNkl[:, k] - logzk[k])
        Nkl_zl = np.exp(logNkl[:, k] - logzk)
        alpha = Nkl_zk + Nkl_zl
        mkmj = m[:, k][:, None] - m
        iCkCj = 1. / (C[:, k][:, None] + C)
        dm[:, k] = (B * Edm - m[:, k] / Ld + (iCkCj * mkmj) @ alpha) / K
        dC[:, k] = (B * EdC - 1. / Ld + (iCkCj - (mkmj * iCkCj) ** 2) @ alpha) / (2 * K)
    sl = slices if isinstance(slices, (list, tuple)) else [slices]
    dL = [-0.5 * (((m[s] ** 2 + C[s]) * iL[s] ** 2).sum() / K - iL[s].sum()) for s in sl]
    dbpars = [-(EdPhi * dP).sum() for dP in dPhis]
    elbo = -np.inf
    if calc_ll:
        elbo = (Ell.sum() * B - 0.5 * D * K * np.log(2 * np.pi) - 0.5 * K * np.log(Ld).sum()
                - 0.5 * ((m ** 2 + C) * iL).sum() - logzk.sum() + np.log(K)) / K
    return -elbo, [-dm, -dC, dL, dlpars, dbpars]


def sgd_update(name, state, x, grad, **hp):
    """One step of the updaters of optimize/sgd.py:14-223; `state` is a dict carried between calls."""
    if name == "sgd":
        return x - hp.get("eta", 0.1) * grad
    if name == "adadelta":
        rho, eps = hp.get("rho", 0.1), hp.get("epsilon", 1e-5)
        state["Eg2"] = rho * state.get("Eg2", 0) + (1 - rho) * grad ** 2
        dx = -grad * np.sqrt(state.get("Edx2", 0) + eps) / np.sqrt(state["Eg2"] + eps)
        state["Edx2"] = rho * state.get("Edx2", 0) + (1 - rho) * dx ** 2
        return x + dx
    if name == "adagrad":
        eta, eps = hp.get("eta", 1), hp.get("epsilon", 1e-6)
        state["g2"] = state.get("g2", 0) + grad ** 2
        return x - eta * grad / (eps + np.sqrt(state["g2"]))
    if name == "momentum":
        rho, eta = hp.get("rho", 0.5), hp.get("eta", 0.01)
        state["dx"] = rho * state.get("dx", 0) - eta * grad
        return x + state["dx"]
    if name == "adam":
        a, b1, b2, eps = hp.get("alpha", 0.01), hp.get("beta1", 0.9), hp.get("beta2", 0.99), hp.get("epsilon", 1e-8)
        state["t"] = state.get("t", 0) + 1
        state["m"] = b1 * state.get("m", 0) + (1 - b1) * grad
        state["v"] = b2 * state.get("v", 0) + (1 - b2) * grad ** 2
        mbar = state["m"] / (1 - b1 ** state["t"])
        vbar = state["v"] / (1 - b2 ** state["t"])
        return x - a * mbar / (np.sqrt(vbar) + eps)
    raise ValueError(name)


# --------------------------------------------------------------------------
# a-15 / f-3  GeneralizedLinearModel.fit: the optimiser stack around `_elbo`
# --------------------------------------------------------------------------

class ParamSpec(object):
    """What the optimiser stack needs of a btypes.Parameter (btypes.py:193-349): a value OR a scipy.stats frozen
    distribution, whether its bound is Positive (log trick) and the bound's limits."""

    def __init__(self, value=None, dist=None, positive=False, lower=None, upper=None, shape=()):
        self.value, self.dist, self.positive = value, dist, positive
        self.lower = 1e-14 if (positive and lower is None) else lower      # btypes.py:176 (Positive's lower)
        self.upper = upper
        self.shape = tuple(shape) if dist is not None else np.shape(value if value is not None else [])

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=int))

    def rvs(self, rs):
        """btypes.py:290-324: a clipped draw, or the value."""
        if self.dist is None:
            return self.value
        s = self.dist.rvs(size=self.shape, random_state=rs)
        if not self.lower and not self.upper:                               # btypes.py:61-79 (clip)
            return s
        return np.clip(s, self.lower, self.upper)


def glm_features(children):
    """The concatenated feature map of a GLM fit as a closure ``(X, [basis parameters]) -> (Phi, [dPhi slab per flat
    length-scale coordinate], [column slice per child])``.  children: ("linear", onescol) | ("rff", W, n_ls) with
    n_ls = 1 (isotropic: the dimension-0 slab only, basis_functions.py:896) or d (ARD).  Gradient slabs are zero-padded
    to the concatenation's width (basis_functions.py:1629-1677)."""
    def build(X, bpars):
        cols, slabs, o = [], [], 0
        for c, p in zip(children, bpars):
            cols.append(linear_transform(X, c[1]) if c[0] == "linear" else rff_transform(X, c[1], p))
        Phi = np.hstack(cols)
        slices = []
        for c, p, blk in zip(children, bpars, cols):
            w = blk.shape[1]
            slices.append(slice(o, o + w))
            if c[0] == "rff":
                g = rff_grad(X, c[1], p)
                for s in ([g] if g.ndim == 2 else [g[:, :, i] for i in range(g.shape[2])]):
                    full = np.zeros_like(Phi)
                    full[:, o:o + w] = s
                    slabs.append(full)
            o += w
        return Phi, slabs, slices
    return build


def glm_fit(X, y, lik, largs, children, regs, likpar, lss, K, L, batch_size, maxiter, nstarts, seed, global_seed,
            sgd_batch_size=10, updater="adam", updater_hp=None):
    """GeneralizedLinearModel.fit (glm.py:141-203) with the optimiser it builds, ``structured_sgd(logtrick_sgd(sgd))``
    (glm.py:176), restated as one loop:

    * structured_sgd (optimize/decorators.py:133-252): the start point is a draw of every random Parameter from NumPy's
      GLOBAL stream (``flatten(..., ravel=bt.ravel)`` -> ``rvs(random_state=None)``, btypes.py:351-371); with nstarts > 0
      the best of nstarts candidates replaces it -- per candidate: a minibatch of ``batch_size`` rows off
      ``gen_batch`` (one RandomState: the estimator's ``random_``), then a draw of every Parameter from that same
      RandomState, then ``_elbo`` (which draws K x randn(L, D) from it) -- first minimum wins (decorators.py:541-583);
    * the main loop's batch size is ``sgd``'s own default of 10: structured_sgd does NOT forward batch_size
      (decorators.py:244-246) -- `sgd_batch_size` (pass batch_size for an implementation that forwards it);
    * logtrick_sgd (decorators.py:329-408, 586-616): z = log x on Positive coordinates, gradient times exp(z), bounds
      (log 1e-100, log upper | log sqrt(max float));
    * sgd (optimize/sgd.py:337-425): fresh ``endless_permutations`` (utils/rand.py:7-31), per step a batch, ``fun``,
      ||grad||, outward gradients truncated on coordinates AT a bound, the updater (Adam by default, sgd.py:14-330), clip;
    * `_elbo`'s iteration counter starts at -nstarts; the objective is only evaluated while it is negative, every 500th
      iteration and at maxiter - 1 (glm.py:232-236), inf otherwise.

    regs / lss: one ParamSpec per child (shape (0,) for a child without length scale); likpar: [] or [ParamSpec].
    Returns (m, C, regs, likpars, lss, objs, norms, next randn of the estimator's RandomState)."""
    from scipy.stats import gamma as _gamma, norm as _norm
    rs = np.random.RandomState(seed)
    grs = np.random.RandomState(global_seed)
    X, y = np.asarray(X, float), np.asarray(y, float)
    N = len(X)
    B = N / batch_size
    feats = glm_features(children)
    D = feats(X[:1], [np.ones(p.shape) if p.size else [] for p in lss])[0].shape[1]
    specs = [ParamSpec(dist=_norm(), shape=(D, K)), ParamSpec(dist=_gamma(a=2, scale=0.5), positive=True, shape=(D, K))] \
        + list(regs) + list(likpar) + list(lss)
    nreg, nlik = len(regs), len(likpar)

    def flat(vals):
        return np.concatenate([np.ravel(np.asarray(v, float)) for v in vals]) if vals else np.empty(0)

    def unflat(x):
        out, o = [], 0
        for p in specs:
            v = x[o:o + p.size]
            out.append(float(v[0]) if p.shape == () else v.reshape(p.shape))
            o += p.size
        return out

    it = [-nstarts]

    def elbo(vals, idx):
        m, C = vals[0], vals[1]
        rg, lp, bp = vals[2:2 + nreg], vals[2 + nreg:2 + nreg + nlik], vals[2 + nreg + nlik:]
        Phi, slabs, slices = feats(X[idx], bp)
        diag = np.concatenate([np.full(s.stop - s.start, r) for s, r in zip(slices, rg)])
        e = np.stack([rs.randn(L, D) for _ in range(K)])
        dolog = (it[0] % 500 == 0) or (it[0] == maxiter - 1)
        calc = dolog or it[0] < 0
        it[0] += 1
        o, (ndm, ndC, dL, dlp, dbp) = glm_elbo(m, C, diag, slices, lik, list(lp), [a[idx] for a in largs], Phi, slabs,
                                                y[idx], e, B, calc_ll=calc)
        return o, flat([ndm, ndC] + list(dL) + list(dlp) + list(dbp))

    def batches(bs):
        perm, pos = np.empty(0, dtype=int), 0
        while True:
            ind = []
            for _ in range(min(bs, N)):
                if pos == len(perm):
                    perm, pos = rs.permutation(N), 0
                ind.append(perm[pos])
                pos += 1
            yield np.array(ind)

    x0 = flat([p.rvs(grs) for p in specs])
    if nstarts > 0:
        gen, best = batches(batch_size), None
        for _ in range(nstarts):
            idx = next(gen)
            cand = [p.rvs(rs) for p in specs]
            obj = elbo(cand, idx)[0]
            if best is None or obj < best[0]:
                best = (obj, cand)
        x0 = flat(best[1])

    pos = np.concatenate([np.full(p.size, p.positive) for p in specs])
    lo = np.concatenate([np.full(p.size, np.log(1e-100) if p.positive else (-np.inf if p.lower is None else p.lower))
                         for p in specs])
    hi = np.concatenate([np.full(p.size, (np.log(np.sqrt(np.finfo(float).max)) if p.upper is None else np.log(p.upper))
                                 if p.positive else (np.inf if p.upper is None else p.upper)) for p in specs])
    z = np.array(x0, dtype=float)
    z[pos] = np.log(z[pos])
    state, objs, norms = {}, [], []
    gen = batches(sgd_batch_size)
    hp = dict(updater_hp or {})
    for _ in range(maxiter):
        idx = next(gen)
        x = np.where(pos, np.exp(np.where(pos, z, 0.)), z)
        obj, g = elbo(unflat(x), idx)
        g = np.where(pos, g * np.exp(np.where(pos, z, 0.)), g)
        objs.append(obj)
        norms.append(np.linalg.norm(g))
        g[z <= lo] = np.minimum(g[z <= lo], 0)
        g[z >= hi] = np.maximum(g[z >= hi], 0)
        z = np.clip(sgd_update(updater, state, z, g, **hp), lo, hi)
    vals = unflat(np.where(pos, np.exp(np.where(pos, z, 0.)), z))
    return (vals[0], vals[1], vals[2:2 + nreg], vals[2 + nreg:2 + nreg + nlik], vals[2 + nreg + nlik:],
            np.array(objs), np.array(norms), rs.randn())


def lik_Ey(name, f, *args):
    """Expected target given the latent function: likelihoods.py Bernoulli :69-84, Binomial :194-211, Gaussian :325-344,
    Poisson :483-498."""
    f = np.asarray(f, float)
    if name == "bernoulli":
        return _expit(f)
    if name == "binomial":
        return _expit(f) * np.asarray(args[0], float)
    if name == "gaussian":
        return f
    if name == "poisson_exp":
        return np.exp(f)
    if name == "poisson_softplus":
        return softplus(f)
    raise ValueError(name)


def lik_cdf(name, y, f, *args):
    """Cumulative distribution of the target: likelihoods.py :133-150, :241-258, :398-423, :523-545 (scipy.stats' cdfs)."""
    from scipy.stats import bernoulli, binom, norm, poisson
    if name == "bernoulli":
        return bernoulli.cdf(y, _expit(f))
    if name == "binomial":
        return binom.cdf(y, n=args[0], p=_expit(f))
    if name == "gaussian":
        return norm.cdf(y, loc=f, scale=np.sqrt(args[0]))
    if name == "poisson_exp":
        return poisson.cdf(y, mu=np.exp(f))
    if name == "poisson_softplus":
        return poisson.cdf(y, mu=softplus(f))
    raise ValueError(name)


def glm_predictions(X, m, C, children, bpars, lik, lpars, largs, nsamples, seed, yq, q):
    """GeneralizedLinearModel's Monte-Carlo predictions (glm.py:351-495, 572-620), every call on a RandomState freshly
    seeded with `seed`: the draws are `randint(0, K, nsamples)` then `randn(D, nsamples)`; latent samples f_s = Phi w_s;
    moments of E[y | f_s], mean / min / max over the samples of the log-density of yq and of the CDF at q."""
    D, K = m.shape
    rs = np.random.RandomState(seed)
    k = rs.randint(0, K, size=(nsamples,))
    w = m[:, k] + rs.randn(D, nsamples) * np.sqrt(C[:, k])
    Phi = glm_features(children)(np.asarray(X, float), bpars)[0]
    fs = (Phi @ w).T                                                                      # (S, N)
    pars = list(lpars) + list(largs)
    ys = np.stack([lik_Ey(lik, f, *pars) for f in fs], axis=1)                            # (N, S)
    Ey = ys.mean(axis=1)
    Vy = ((ys - Ey[:, None]) ** 2).mean(axis=1)
    lp = np.stack([lik_loglike(lik, yq, f, *pars) for f in fs], axis=1)
    cd = np.stack([lik_cdf(lik, q, f, *pars) for f in fs], axis=1)
    return {"fs": fs, "Ey": Ey, "Vy": Vy, "logpdf": (lp.mean(axis=1), lp.min(axis=1), lp.max(axis=1)),
            "cdf": (cd.mean(axis=1), cd.min(axis=1), cd.max(axis=1))}


def slm_random_starts(X, y, W, var_spec, reg_spec, ls_spec, nstarts, seed):
    """The start point of StandardLinearModel.fit (slm.py:112-125) as structured_minimizer picks it (decorators.py:79-97,
    541-583): one draw of (var, regulariser, length scales) from the estimator's RandomState (the flatten's `rvs`), then
    `nstarts` candidates, each scored by `_elbo` (-ELBO); the first minimum wins.  Returns (best candidate, objectives,
    the RandomState's next randn)."""
    rs = np.random.RandomState(seed)
    specs = (var_spec, reg_spec, ls_spec)
    [p.rvs(rs) for p in specs]                       # the start the candidates then replace
    best, objs = None, []
    for _ in range(nstarts):
        var, reg, ls = [p.rvs(rs) for p in specs]
        Phi = rff_transform(X, W, ls)
        g = rff_grad(X, W, ls)
        slabs = [g] if g.ndim == 2 else [g[:, :, i] for i in range(g.shape[2])]
        obj = -slm_elbo(Phi, y, var, np.full(Phi.shape[1], reg), slice(None), slabs)["elbo"]
        objs.append(obj)
        if best is None or obj < best[0]:
            best = (obj, (var, reg, ls))
    return best[1], np.array(objs), rs.randn()
