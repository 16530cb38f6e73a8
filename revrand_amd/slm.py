"""
The Bayesian standard linear model with revrand's interface (reference: revrand/slm.py), its
O(N F^2) work on the MI355X.

Same constructor, ``fit`` / ``predict`` / ``predict_moments`` and fitted attributes
(``var_, regularizer_, hypers_, weights_, covariance_, obj_``) as
``revrand.slm.StandardLinearModel``; sklearn-clonable.  Inside ``_elbo`` (slm.py:142-199):

* ``Phi = basis.transform(X)``                 -> HIP (random bases)               slm.py:145
* ``Phi.T.dot(Phi)``, ``Phi.T.dot(y)``         -> ``basis.gram`` (fused, Phi never built) for a
  single random Fourier basis, else ``rr_dense_gram`` (same MFMA SYRK kernel)       slm.py:146,157
* ``(dPhi.T.dot(Phi) * C).sum()``              -> the (Phi, dPhi) block of ``rr_dense_gram([Phi | dPhi])``
                                                                                   slm.py:195
* Cholesky / inverse ``solve_posdef``          -> host, by design                   slm.py:155

Everything else is O(N F) or O(F^2) host arithmetic exactly as in the reference.
"""
import logging
import os
from functools import partial

import numpy as np
from scipy.optimize import minimize
from scipy.stats import gamma
from sklearn.base import BaseEstimator, RegressorMixin
from sklearn.utils import check_random_state
from sklearn.utils.validation import check_array, check_is_fitted, check_X_y

from . import _hip
from .basis_functions import LinearBasis, apply_grad
from .btypes import Parameter, Positive
from .linalg import solve_posdef
from .optimize import logtrick_minimizer, structured_minimizer
from .utils import atleast_list, issequence

log = logging.getLogger(__name__)

import inspect as _inspect
# sklearn renamed check_array's `force_all_finite` to `ensure_all_finite` (1.6)
_NO_FINITE_CHECK = {("ensure_all_finite" if "ensure_all_finite" in _inspect.signature(check_array).parameters
                     else "force_all_finite"): False}


def _scale_nested(g, s):
    """Every leaf of a nested gradient structure times s."""
    if isinstance(g, (list, tuple)):
        return [_scale_nested(u, s) for u in g]
    return np.asarray(g, dtype=float) * s if np.ndim(g) else float(g) * s


class StandardLinearModel(BaseEstimator, RegressorMixin):
    """Bayesian linear regression on a basis; hyper-parameters by L-BFGS-B on the ELBO.

    Parameters are those of the reference (slm.py:57-72): ``basis``, ``var``, ``tol``,
    ``maxiter``, ``nstarts``, ``random_state``, plus

    distributed : bool
        Row-sharded fit, one process per GPU (``revrand_amd.parallel``: RCCL bound directly, ranks from the
        launcher's environment): every rank calls ``fit`` with ITS rows; each ``_elbo`` sums the per-rank
        statistics ``[tri G | b | y^T y | N]`` and ``[sqErr | dhyp]`` with one all-reduce each, so all ranks walk the same
        L-BFGS path and end with identical parameters.  Needs a single random-feature basis (the
        device-resident path) and a fixed ``random_state`` shared by all ranks when ``nstarts > 0``.
    devices : None | sequence of GPU indices | int | "all"
        Several GPUs behind THIS call, in this process (``revrand_amd.multigpu``): the rows of ``fit`` / ``predict_moments``
        are sharded over the listed GPUs (an int n: GPUs 0..n-1; "all": every visible one; an index may repeat), the
        per-GPU statistics are summed in HBM by one in-process collective (RCCL, or the library's peer kernels over xGMI)
        and the posterior is replicated.  No launcher, no environment: what a ``Pipeline`` / ``GridSearchCV`` caller of the
        reference (one ``fit`` call, slm.py:74-140) can use.  None: one GPU (``REVRAND_HIP_DEVICE`` / 0).
    gram_engine : None | "f32" | "fp16x3" | "bf16x3" | "bf16x4"
        Arithmetic of ``Phi^T Phi`` and ``Phi C`` for "f32" bases during ``fit`` / ``predict_moments``
        (``include/revrand_hip.h``, RR_GRAM_*): None keeps the device context's setting (exact f32 MFMA unless
        ``RR_SYRK_ENGINE`` says otherwise); "fp16x3" is ~2.5x faster at the f32 engine's accuracy.
    """

    def __init__(self, basis=LinearBasis(), var=Parameter(gamma(1.), Positive()), tol=1e-8, maxiter=1000,
                 nstarts=100, random_state=None, distributed=False, gram_engine=None, devices=None):
        self.basis = basis
        self.gram_engine = gram_engine
        self.devices = devices
        self.var = var
        self.tol = tol
        self.maxiter = maxiter
        self.nstarts = nstarts
        self.random_state = random_state
        self.distributed = distributed
        self.random_ = check_random_state(random_state)

    def fit(self, X, y):
        """Learn (var, regularizer, basis hyper-parameters); returns self (slm.py:74-140)."""
        with self._engine_scope():
            return self._fit(X, y)

    def _group(self):
        """The device group of ``devices=`` (None: the single default device)."""
        if getattr(self, "devices", None) is None:
            return None
        from . import multigpu
        return multigpu.get_group(self.devices)

    def _engine_scope(self):
        g = self._group()
        return _hip.gram_engine_scope(getattr(self, "gram_engine", None), None if g is None else g.members)

    def _fit(self, X, y):
        X, y = check_X_y(X, y)
        self._drop_serving()
        self.obj_ = -np.inf
        params = [self.var, self.basis.regularizer, self.basis.params]
        nmin = structured_minimizer(logtrick_minimizer(minimize))
        inner = partial(StandardLinearModel._elbo, self, X, y)

        def elbo(*values):   # (notes the first point evaluated WITH a gradient: the optimiser's start, after the random starts)
            if elbo.first_point is None:
                from .utils import flatten_values
                elbo.first_point = np.asarray(flatten_values(list(values)), dtype=float)
            return inner(*values)
        elbo.first_point = None
        elbo.objective_only = partial(StandardLinearModel._elbo_objective, self, X, y)  # used by the random starts
        # a single random-feature basis keeps (X, y) on the GPU for the whole optimisation
        self._state = self._make_state(X, y)
        if self.distributed and self._state is None:
            raise ValueError("distributed=True needs random-feature bases in f32 mode (alone or concatenated "
                             "with Linear/Bias bases)")
        self._defer_cov = True
        try:
            res = nmin(elbo, params, method="L-BFGS-B", jac=True, tol=self.tol,
                       options={"maxiter": self.maxiter, "maxcor": 100}, random_state=self.random_,
                       nstarts=self.nstarts)
            res = self._retry_stalled_start(res, nmin, elbo, params)
            if self._state is not None and getattr(self._state, "best_on_device", False):
                self.covariance_ = self._state.best_covariance()
            if self.distributed and not self._ranks_bit_identical(getattr(self._state, "best_on_device", False)):
                # posteriors agree to the last bits only (see `_same_on_all_ranks`): ship rank 0's
                from . import parallel
                comm = parallel.get_comm()
                self.weights_ = comm.broadcast_host(np.ascontiguousarray(self.weights_, dtype=np.float64), root=0)
                self.covariance_ = comm.broadcast_host(np.ascontiguousarray(self.covariance_, dtype=np.float64), root=0)
        finally:
            self._defer_cov = False
            if self._state is not None:
                self._state.release()
            self._state = None
        self.var_, self.regularizer_, self.hypers_ = res.x
        log.info("Done! ELBO = {}, var = {}, reg = {}, hypers = {}, message = {}."
                 .format(-res["fun"], self.var_, self.regularizer_, self.hypers_, res.message))
        return self

    def _retry_stalled_start(self, res, nmin, elbo, params):
        """L-BFGS-B's first trial point is x0 - g (log space for Positive parameters): with |g| in the hundreds that is
        var = 1e-93 and an objective of 1e94, from which its line search interpolates to a step that rounds to ZERO -- it then
        reports "convergence" at the start point after no iteration (scipy: REL_REDUCTION_OF_F <= FACTR*EPSMCH), or walks
        on, depending on the last bits of that absurd value (seen with float32 statistics: two kernels agreeing to 2e-8
        sent the same fit either way; the reference's float64 run is subject to the same chance).  A fit that made NO
        iteration from a point that is not stationary is retried ONCE from the same point with objective and gradient
        scaled by 1 / max|g|: the first trial step is then of unit length.  Fits that iterate are untouched."""
        from .utils import flatten_values
        first = getattr(elbo, "first_point", None)   # (the wrapper below notes where the optimiser started)
        moved = first is None or not np.array_equal(np.asarray(flatten_values(list(res.x)), dtype=float), first)
        if getattr(res, "nit", 2) > 1 or moved or self.maxiter < 1 or not np.isfinite(res["fun"]):
            return res
        g0 = np.abs(np.asarray(flatten_values(list(res.jac)), dtype=float)) if "jac" in res else np.zeros(1)
        gmax = float(g0.max()) if g0.size else 0.0
        if not np.isfinite(gmax) or gmax <= 10.0:
            return res
        scale = 1.0 / gmax

        def scaled(*a, **k):
            f, g = elbo(*a, **k)
            return f * scale, _scale_nested(g, scale)
        log.info("No iteration from a non-stationary start (max |gradient| = %g): retrying with a scaled objective", gmax)
        obj_keep = self.obj_
        res2 = nmin(scaled, params, method="L-BFGS-B", jac=True, tol=self.tol,
                    options={"maxiter": self.maxiter, "maxcor": 100}, random_state=self.random_, nstarts=0,
                    start_values=list(res.x))
        if res2["fun"] / scale < res["fun"]:
            res2["fun"] = res2["fun"] / scale
            return res2
        self.obj_ = obj_keep
        return res

    def _make_state(self, X, y):
        """(X, y) resident on the device for the whole optimisation, when the basis supports it."""
        group = self._group()
        if group is not None:
            if self.distributed:
                raise ValueError("devices= (several GPUs in this process) and distributed=True (one process per GPU) "
                                 "cannot be combined")
            from . import multigpu
            st = multigpu.ShardedFitState.make(self.basis, X, y, group)
            if st is not None:
                return st
            log.info("devices=%s: this basis / row count has no sharded device-resident fit; one GPU is used", self.devices)
        make = getattr(self.basis, "device_fit_state", None)
        return make(X, y) if make is not None else None

    def _allreduce(self, buf):
        """Sum a float64 vector over the ranks (no-op unless distributed)."""
        if not self.distributed:
            return buf
        from . import parallel
        return parallel.allreduce_host(buf)

    # -- device statistics -------------------------------------------------------------------
    def _gram(self, X, y, Phi, hyp):
        """(Phi^T Phi, Phi^T y) on the GPU: fused when the basis offers it, dense SYRK otherwise."""
        gram = getattr(self.basis, "gram", None)
        res = gram(X, y, *hyp) if gram is not None else None
        if res is None:  # no fused route (or it declined: float64 children): Gram of Phi in Phi's own dtype
            res = _hip.dense_gram(Phi, y)
        return res[0], res[1]

    @staticmethod
    def _cross_gram(Phi, dPhi):
        """Phi^T dPhi through the SYRK kernel: the off-diagonal block of [Phi | dPhi]^T [Phi | dPhi]."""
        F = Phi.shape[1]
        G2, _, _ = _hip.dense_gram(np.hstack((Phi, dPhi)))
        return G2[:F, F:]

    def _elbo_resident(self, X, y, var, reg, hypers, objective_only=False):
        """`_elbo` with (X, y) resident on the device: two data passes on the GPU (statistics; Err / U = Phi C /
        gradient contraction) around the posterior -- Cholesky, inverse and the O(F^2) reductions also in HBM
        (rr_posterior_dev) unless the statistics are summed over ranks or the matrix needs the SVD route; neither
        Phi nor dPhi ever materialised, O(F) numbers over PCIe per evaluation."""
        st = self._state
        N = X.shape[0]
        # posterior on the device (Cholesky + inverse + reductions in HBM) for F >= 256 unless RR_POSDEF=host or the
        # ranks' statistics cannot be summed in HBM (no RCCL group: gloo / CPU tests)
        on_dev = hasattr(st, "posterior") and _hip.posterior_available(getattr(st, "F", None))
        if self.distributed:
            from . import parallel
            comm = parallel.get_comm()
            on_dev = on_dev and comm.device_reduce
        if on_dev:
            # distributed: the one exchange -- [tri G | b | y^T y | N] of all row shards summed in HBM over RCCL / xGMI
            # (y^T y is read by the objective-only evaluation alone: a full `_elbo` takes sqErr from its second pass and leaves
            # it in HBM -- one host round trip less per evaluation; the host SVD route fetches all statistics anyway)
            lazy = not objective_only and not self.distributed and getattr(st, "lazy_yty", False)
            yty = st.gram_device(hypers, comm.reduce_stats_device if self.distributed else None, **({"want_yty": False} if lazy else {}))
            if self.distributed:
                N = st.N_total
            D = st.F
        else:
            PhiPhi, Phiy, yty = st.gram(hypers)
            D = PhiPhi.shape[0]
            if self.distributed:  # one exchange: the packed sufficient statistics of all row shards
                PhiPhi, Phiy, yty, N = parallel.unpack_stats(
                    self._allreduce(parallel.pack_stats(PhiPhi, Phiy, yty, N)), D)
        L, slices = self.basis.regularizer_diagonal(X, *atleast_list(reg))
        iL = 1. / L
        post = st.posterior(iL, var) if on_dev else None
        if post is None:
            if on_dev:  # not safely positive definite: the reference's SVD route, on the host
                log.info("posterior not safely positive definite on the device: host solve_posdef for this step")
                PhiPhi, Phiy, yty = st.stats_host()
            iC = np.diag(iL) + PhiPhi / var
            C, logdetiC = solve_posdef(iC, np.eye(D))
            m = C.dot(Phiy) / var
            TrPhiPhiC = (PhiPhi * C).sum()
            Cdiag, Cpass = C.diagonal(), C
        else:
            m, Cdiag, logdetiC, TrPhiPhiC = post
            Cpass = st.dC
        logdetC = -logdetiC
        if objective_only:
            # -ELBO alone needs no second data pass: sqErr = y^T y - 2 m^T b + m^T G m and, because
            # (diag(iL) + G / var) m = b / var,  m^T G m = m^T b - var sum(iL m^2).  (The statistics are already summed
            # over the ranks.)  No side effects: the optimiser re-evaluates its start point with `_elbo`.
            bvec = Phiy if post is None else st.b_host()
            sqErr = yty - m.dot(bvec) - var * ((m ** 2) * iL).sum()
            ELBO = -0.5 * (N * np.log(2 * np.pi * var) + sqErr / var + TrPhiPhiC / var
                           + ((m ** 2 + Cdiag) * iL).sum() - logdetC + np.log(L).sum() - D)
            if self.distributed and not self._ranks_bit_identical(post is not None):
                ELBO = float(comm.broadcast_host(np.array([ELBO]), root=0)[0])  # see `_same_on_all_ranks`
            return -ELBO
        sqErr, dhypers = st.second_pass(hypers, m, Cpass, var)
        if self.distributed:  # second exchange: 1 + (number of length scales) numbers
            parts = dhypers if isinstance(dhypers, list) else [dhypers]
            red = self._allreduce(np.concatenate([[sqErr]] + [np.atleast_1d(p) for p in parts]))
            sqErr, out, k = float(red[0]), [], 1
            for p in parts:
                out.append(float(red[k]) if np.ndim(p) == 0 else red[k:k + np.size(p)])
                k += np.size(p)
            dhypers = out if isinstance(dhypers, list) else out[0]
        ELBO = -0.5 * (N * np.log(2 * np.pi * var) + sqErr / var + TrPhiPhiC / var
                       + ((m ** 2 + Cdiag) * iL).sum() - logdetC + np.log(L).sum() - D)
        dvar = 0.5 * (-N + (sqErr + TrPhiPhiC) / var) / var

        def dreg(s):
            return -0.5 * (((m[s] ** 2 + Cdiag[s]) * iL[s] ** 2).sum() - iL[s].sum())

        dL = list(map(dreg, slices)) if issequence(slices) else dreg(slices)
        if self.distributed and not self._ranks_bit_identical(post is not None):
            ELBO, dvar, dL, dhypers = self._same_on_all_ranks(comm, [ELBO, dvar, dL, dhypers])
        if ELBO > self.obj_:
            self.weights_ = m
            self.obj_ = ELBO
            if post is None:
                self.covariance_ = C
                st.best_on_device = False
            else:
                st.keep_best()  # stays in HBM; fetched once at the end of fit
                if not getattr(self, "_defer_cov", False):
                    self.covariance_ = st.best_covariance()
        if log.isEnabledFor(logging.INFO):   # (the reference's line, slm.py:179-180; formatted only when somebody listens)
            log.info("ELBO = {}, var = {}, reg = {}, hypers = {}.".format(ELBO, var, reg, hypers))
        return -ELBO, [-dvar, dL, dhypers]

    def _ranks_bit_identical(self, posterior_on_device):
        """In deterministic mode (Device.set_deterministic / RR_DETERMINISTIC=1) the posterior kernels and every device
        reduction sum in a fixed order: ranks holding the same all-reduced statistics compute the same bits, and the
        evaluation needs its two exchanges only ([tri G | b | y^T y | N], then [sqErr | dhyp]) -- no broadcast of the results.
        That claim covers the DEVICE kernels only, and only when EVERY rank runs them that way:
        * the flag is per process (an environment variable), so the ranks agree on it ONCE per fit state -- an all-reduce
          (min) of the flag, cached on the state -- instead of each trusting its own: a rank that skipped a broadcast its
          peers entered would hang the job in the next collective;
        * a step whose posterior came from the HOST route (solve_posdef / LAPACK after RR_ERR_NOT_POSDEF, or RR_POSDEF=host)
          keeps the broadcast: different CPUs / BLAS builds need not agree to the last bit.  (With identical statistics and
          deterministic device kernels every rank takes the host route on the same steps, so the ranks still agree on
          WHETHER to broadcast.)"""
        st = getattr(self, "_state", None)
        dev = getattr(st, "dev", None)  # the fit state's device context, if it has one
        if st is None or dev is None:
            return False
        agreed = getattr(st, "_deterministic_on_all_ranks", None)
        if agreed is None:
            from . import parallel
            mine = 1.0 if getattr(dev, "deterministic", False) else 0.0
            agreed = st._deterministic_on_all_ranks = bool(parallel.get_comm().allreduce_host(np.array([mine]), op="min")[0] > 0.5)
        return bool(agreed and posterior_on_device)

    @staticmethod
    def _same_on_all_ranks(comm, values):
        """Rank 0's numbers for every rank.  The ranks hold identical reduced statistics, but the posterior kernels sum
        with f64 atomics whose order differs from run to run, so objective and gradients agree to the last bits only;
        L-BFGS-B's line-search decisions must not depend on those bits (ranks taking different numbers of evaluations
        would leave each other waiting in the next collective).  One broadcast of 3 + d numbers per evaluation."""
        from .utils import flatten_values, shapes_of, unflatten
        flat = comm.broadcast_host(flatten_values(values), root=0)
        return unflatten(flat, shapes_of(values))

    def _elbo_objective(self, X, y, var, reg, hypers):
        """-ELBO only (what the random starts of `fit` rank by, decorators.py:541-583): with the data resident on the
        device this costs one statistics pass + the posterior, a third of a full `_elbo` at F = 4096."""
        if getattr(self, "_state", None) is not None:
            return self._elbo_resident(X, y, var, reg, hypers, objective_only=True)
        return self._elbo(X, y, var, reg, hypers)[0]

    def _elbo(self, X, y, var, reg, hypers):
        if getattr(self, "_state", None) is not None:
            return self._elbo_resident(X, y, var, reg, hypers)
        hyp = atleast_list(hypers)
        Phi = self.basis.transform(X, *hyp)  # N x D
        N, D = Phi.shape
        PhiPhi, Phiy = self._gram(X, y, Phi, hyp)

        L, slices = self.basis.regularizer_diagonal(X, *atleast_list(reg))
        iL = 1. / L

        # posterior (host Cholesky)
        iC = np.diag(iL) + PhiPhi / var
        C, logdetiC = solve_posdef(iC, np.eye(D))
        logdetC = -logdetiC
        m = C.dot(Phiy) / var

        TrPhiPhiC = (PhiPhi * C).sum()
        Err = y - Phi.dot(m)
        sqErr = (Err ** 2).sum()

        ELBO = -0.5 * (N * np.log(2 * np.pi * var) + sqErr / var + TrPhiPhiC / var
                       + ((m ** 2 + C.diagonal()) * iL).sum() - logdetC + np.log(L).sum() - D)

        # keep the best posterior seen (slm.py:173-177)
        if ELBO > self.obj_:
            self.weights_ = m
            self.covariance_ = C
            self.obj_ = ELBO

        if log.isEnabledFor(logging.INFO):   # (the reference's line, slm.py:179-180; formatted only when somebody listens)
            log.info("ELBO = {}, var = {}, reg = {}, hypers = {}.".format(ELBO, var, reg, hypers))

        dvar = 0.5 * (-N + (sqErr + TrPhiPhiC) / var) / var

        def dreg(s):
            return -0.5 * (((m[s] ** 2 + C[s, s].diagonal()) * iL[s] ** 2).sum() - iL[s].sum())

        dL = list(map(dreg, slices)) if issequence(slices) else dreg(slices)

        def dhyps(dPhi):
            # slm.py:193-195 with (dPhi^T Phi o C).sum() == (Phi^T dPhi o C).sum() (C symmetric)
            return -(m.T.dot(Err.dot(dPhi)) - (self._cross_gram(Phi, dPhi) * C).sum()) / var

        dhypers = apply_grad(dhyps, self.basis.grad(X, *hyp))

        return -ELBO, [-dvar, dL, dhypers]

    # -- serving state: what repeated predictions reuse between calls ---------------------------------
    def _serving(self):
        """Per-process device state of a fitted estimator: the posterior covariance in HBM (uploaded once instead of
        converted and copied per `predict_moments` call: 9 ms at F = 4096) and the feature matrix of `predict`.
        Dropped by `fit`, by pickling and when `covariance_` is replaced."""
        C = self.covariance_
        key = (id(C), np.shape(C), float(np.asarray(C).flat[0]), float(np.asarray(C).flat[-1]))
        s = self.__dict__.get("_serve")
        if s is None or s["pid"] != os.getpid() or s["key"] != key:
            self._drop_serving()
            s = self.__dict__["_serve"] = {"pid": os.getpid(), "key": key, "cov": {}, "feats": None}
        return s

    def _drop_serving(self):
        s = self.__dict__.pop("_serve", None)
        if s is not None and s["pid"] == os.getpid():
            if s["feats"] is not None:
                s["feats"].release()
            for cov in s["cov"].values():
                cov.free()

    def _device_covariance(self):
        """The posterior covariance in the HBM of the calling thread's device (uploaded once per device context)."""
        s = self._serving()
        key = _hip.device_key()
        if key not in s["cov"]:
            s["cov"][key] = _hip.DeviceCovariance(_hip.get_device(), self.covariance_)
        return s["cov"][key]

    def __getstate__(self):
        state = dict(super().__getstate__())  # sklearn's (adds its version tag)
        state.pop("_serve", None)
        return state

    def predict(self, X):
        """Predictive mean (slm.py:201-217).  The reference computes it through `predict_moments`, i.e. with the
        N x F x F product of the variance; the mean alone is Phi m: for a random kernel basis straight out of the feature
        kernel (rr_rff_predict_mean_dev: no product at all, no feature matrix in HBM, the rows validated on a second host
        thread during their upload), for the other f32 bases a dot product per row of the assembled feature matrix
        (rr_featmat_project with one vector); `predict_moments` for bases that can do neither."""
        check_is_fitted(self, ["var_", "regularizer_", "weights_", "covariance_", "hypers_"])
        if getattr(self.basis, "_predict_checks_rows", False) and getattr(self.basis, "predict_moments", None) is not None:
            Xs = check_array(X, **_NO_FINITE_CHECK)
            # the deferred validation sees the caller's WHOLE rows, whatever columns the basis' `apply_ind` keeps (the
            # reference validates X before any slicing, slm.py:214)
            with self._engine_scope():
                res = self._sharded_moments(Xs, None, deferred=True)
            if res is not None:
                return res[0]
        X = check_array(X)
        Ey = self._predict_mean(X)
        if Ey is None:
            Ey, _ = self.predict_moments(X)
        return Ey

    def _predict_mean(self, X):
        bases = getattr(self.basis, "bases", [self.basis])
        if any(getattr(b, "dtype", "f32") != "f32" for b in bases):  # f64 arithmetic was asked for: keep it
            return None
        from .basis_functions import MinibatchFeatures
        srv = self._serving()
        if srv["feats"] is None:
            srv["feats"] = MinibatchFeatures(self.basis)
        with self._engine_scope():
            return srv["feats"].project(X, atleast_list(self.hypers_), np.asarray(self.weights_, dtype=float)[:, None])[:, 0]

    def predict_moments(self, X):
        """Predictive mean and variance (slm.py:219-244)."""
        with self._engine_scope():
            return self._predict_moments(X)

    def _sharded_moments(self, X, with_cov, deferred):
        """``basis.predict_moments`` of the query rows -- on the one default device, or row-sharded over the members of
        ``devices=`` (every member uploads the covariance once and serves its rows; rows are independent, so the result is
        the single-device one bit for bit).  with_cov None: the mean alone.  deferred: the basis validates the rows itself
        (finiteness, under the GPU's work); it is handed a check of the caller's whole rows, not of its column slice."""
        pm = self.basis.predict_moments
        hyp, w = self.hypers_, self.weights_

        def serve(Xr):
            kw = {"check_rows": (lambda _sliced, Xr=Xr: check_array(Xr))} if deferred else {}
            return pm(Xr, hyp, w, self._device_covariance() if with_cov else None, **kw)
        group = self._group()
        if group is None or X.shape[0] < 2 * group.n:
            return serve(X)
        from . import multigpu
        if with_cov:
            self._serving()   # (created on THIS thread: the members' threads then only add their own entry to it)
        return multigpu.map_rows(group, X.shape[0], lambda i, s, e: serve(X[s:e]))

    def _predict_moments(self, X):
        check_is_fitted(self, ["var_", "regularizer_", "weights_", "covariance_", "hypers_"])
        # A basis whose device route validates the rows itself gets them unchecked for finiteness here (shape and dtype only):
        # `check_array` costs as much as the upload of a large query, and runs chunk by chunk under the GPU's work instead
        # (same exception, raised before anything is returned)
        deferred = getattr(self.basis, "_predict_checks_rows", False) and getattr(self.basis, "predict_moments", None) is not None
        X = check_array(X, **(_NO_FINITE_CHECK if deferred else {}))
        if getattr(self.basis, "predict_moments", None) is not None:
            # on the GPU(s), with the covariance already resident there
            res = self._sharded_moments(X, True, deferred)
            if res is not None:
                return res[0], res[1] + self.var_
        if deferred:
            check_array(X)
        # bases without a fused device route (LinearBasis alone, float64 children in a concatenation): their transform,
        # then the N x F x F product on the GPU in float64 (rr_dense_predict) -- never on the host
        Phi = self.basis.transform(X, *atleast_list(self.hypers_))
        Ey, Vf = _hip.dense_predict(Phi, self.weights_, self.covariance_)
        return Ey, Vf + self.var_

    def __repr__(self):
        return "{}(basis={}, var={}, tol={}, maxiter={}, nstarts={}, random_state={})".format(
            type(self).__name__, self.basis, self.var, self.tol, self.maxiter, self.nstarts, self.random_state)
