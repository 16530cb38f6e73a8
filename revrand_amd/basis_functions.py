"""
Basis objects with revrand's Basis protocol, backed by the HIP kernels of librevrand_hip.so.

Same names, constructor arguments, argument meaning, error messages and array shapes as
``revrand.basis_functions`` (reference file cited per item), so the reference's estimators
and these are interchangeable on this path.  What differs is where the arithmetic runs:

* ``transform`` / ``grad`` of the random Fourier bases (RandomRBF, RandomLaplace,
  RandomCauchy, RandomMatern32/52, OrthogonalRBF) call ``rr_rff_transform`` / ``rr_rff_grad``;
* those bases additionally expose ``gram(X, y, *params)`` -- the fused
  Phi -> (Phi^T Phi, Phi^T y, y^T y) accumulation (``rr_rff_gram``) that
  ``StandardLinearModel`` uses so that Phi never has to exist.

Host-side by design (as in the reference): sampling of W from a seeded
``RandomState`` (so seeds reproduce the reference's W bit-for-bit), parameter plumbing,
concatenation bookkeeping.  There is NO NumPy fallback for the device parts: without the
library or a GPU they raise.

New keyword on the HIP-backed bases: ``dtype`` = "f32" (default; the arithmetic type of the
kernels, BASELINE config 2) or "f64".  Returned arrays are always float64, like the
reference's.
"""
import ctypes
import inspect
from functools import reduce, wraps
from itertools import repeat

import threading

import numpy as np
from scipy.linalg import qr
from scipy.stats import gamma, norm as norm_dist
from sklearn.utils import check_random_state

from . import _hip
from .btypes import Bound, Parameter, Positive
from .utils import atleast_list, atleast_tuple, issequence


# --------------------------------------------------------------------------------------
# helpers (reference: basis_functions.py:34-152)
# --------------------------------------------------------------------------------------

def count_args(func):
    """Number of arguments of a (bound) method, excluding self (basis_functions.py:52-66)."""
    return len(inspect.signature(func).parameters)


def slice_init(func):
    """Add the ``apply_ind`` keyword to a basis constructor (basis_functions.py:70-93)."""
    @wraps(func)
    def new_init(self, *args, **kwargs):
        apply_ind = kwargs.pop("apply_ind", None)
        if np.isscalar(apply_ind):
            apply_ind = [apply_ind]
        func(self, *args, **kwargs)
        self.apply_ind = apply_ind
    return new_init


def slice_transform(func):
    """Apply ``X[:, apply_ind]`` before transform/grad/gram (basis_functions.py:96-105).

    ``functools.wraps`` records ``__wrapped__``, which ``inspect.signature`` follows, so
    ``count_args`` still sees the wrapped method's own parameters -- BasisCat routes
    positional hyper-parameters by that count.
    """
    @wraps(func)
    def wrapper(self, X, *vargs, **kwargs):
        if self.apply_ind is not None:
            X = X[:, self.apply_ind]
        return func(self, X, *vargs, **kwargs)
    return wrapper


def apply_grad(fun, grad):
    """Map a functional of a 2-d gradient over structured gradients (basis_functions.py:109-152).

    Sequences/generators recurse (a one-element result is unwrapped), ``[]`` stays ``[]``,
    3-d arrays are mapped over their last axis.
    """
    if issequence(grad):
        fgrad = [apply_grad(fun, g) for g in grad]
        return fgrad if len(fgrad) != 1 else fgrad[0]
    if len(grad) == 0:
        return []
    if grad.ndim in (1, 2):
        return fun(grad)
    if grad.ndim == 3:
        return np.array([fun(grad[:, :, i]) for i in range(grad.shape[2])])
    raise ValueError("Only up to 3d gradients allowed!")


# --------------------------------------------------------------------------------------
# Basis protocol (reference: basis_functions.py:159-384)
# --------------------------------------------------------------------------------------

class Basis(object):
    """Base class: identity transform, no parameters, scalar regulariser."""

    _params = Parameter()
    _regularizer = Parameter(gamma(1.), Positive())

    @slice_init
    def __init__(self, regularizer=None):
        if regularizer is not None:
            if not regularizer.is_scalar:
                raise ValueError("Regularizer parameters have to be scalar!")
            if regularizer.bounds.lower <= 0:
                raise ValueError("Regularizer has to be bounded below by 0!")
            self._regularizer = regularizer

    @slice_transform
    def transform(self, X):
        return X

    @slice_transform
    def grad(self, X):
        return []

    def get_dim(self, X):
        """Output dimensionality, probed once with the first row (basis_functions.py:276-297)."""
        if not hasattr(self, "_D"):
            self._D = self.transform(X[[0]], *self.params_values()).shape[1]
        return self._D

    def params_values(self):
        return [p.value for p in atleast_list(self.params) if p.has_value]

    def regularizer_diagonal(self, X, regularizer=None):
        """(diag of the prior weight variance (D,), slice it applies to)  (:307-340)."""
        reg = self.regularizer.value if regularizer is None else regularizer
        return np.full(self.get_dim(X), reg, dtype=float), slice(None)

    def _transform_popargs(self, X, *args):
        mine, rest = self.__split(args, self.transform)
        return self.transform(X, *mine), rest

    def _put_features_popargs(self, X, fm, col0, *args):
        """Write this basis' columns of a device feature matrix (concatenated Gram); returns the
        parameters left for the following bases."""
        mine, rest = self.__split(args, self.transform)
        self._put_features(X, fm, col0, *mine)
        return rest

    def _put_features(self, X, fm, col0, *params):
        # generic bases: host transform, one upload of their (usually narrow) column block
        fm.put_host(self.transform(X, *params), col0)

    def _resident_child(self, X, dtype=None):
        """This basis' share of a concatenated device-resident fit (CatFitState); None = not supported.  dtype="f64": the
        state's feature matrix is float64 (a child asked for the reference's arithmetic)."""
        return None

    def _grad_popargs(self, X, *args):
        mine, rest = self.__split(args, self.grad)
        return self.grad(X, *mine), rest, mine

    def __split(self, args, fn):
        k = count_args(fn) - 1  # minus X
        return args[:k], args[k:]

    @property
    def params(self):
        return self._params

    @property
    def regularizer(self):
        return self._regularizer

    def __add__(self, other):
        return BasisCat([self, other])

    def __radd__(self, other):
        return self if other == 0 else self.__add__(other)

    def __repr__(self):
        return "{}(regularizer={})".format(type(self).__name__, self.regularizer)


class BiasBasis(Basis):
    """A constant column (basis_functions.py:387-440)."""

    @slice_init
    def __init__(self, offset=1., regularizer=None):
        self.offset = offset
        super(BiasBasis, self).__init__(regularizer)

    @slice_transform
    def transform(self, X):
        return np.ones((len(X), 1)) * self.offset

    def _resident_child(self, X, dtype=None):
        return _ResidentHost(self, X.shape[1])

    def __repr__(self):
        return "{}(offset={}, regularizer={})".format(type(self).__name__, self.offset, self.regularizer)


class LinearBasis(Basis):
    """[1, X] or X (basis_functions.py:443-493); trivially host-side."""

    @slice_init
    def __init__(self, onescol=True, regularizer=None):
        self.onescol = onescol
        super(LinearBasis, self).__init__(regularizer)

    @slice_transform
    def transform(self, X):
        N, D = X.shape
        return np.hstack((np.ones((N, 1)), X)) if self.onescol else X

    @slice_transform
    def _put_features(self, X, fm, col0):
        dX = fm.dev.upload_matrix(np.ascontiguousarray(X, dtype=np.float32))
        fm.put_linear(dX, self.onescol, col0)
        fm.dev.sync()
        dX.free()

    @slice_transform
    def _resident_child(self, X, dtype=None):
        return _ResidentLinear(self, X, dtype)

    def __repr__(self):
        return "{}(onescol={}, regularizer={})".format(type(self).__name__, self.onescol, self.regularizer)


class _LengthScaleBasis(Basis):
    """Length-scale validation shared by the kernel bases (basis_functions.py:579-613)."""

    def _init_lenscale(self, lenscale):
        if (lenscale.shape != (self.d,)) and (lenscale.shape != ()):
            raise ValueError("Parameter dimension doesn't agree with X dimensions!")
        self._params = lenscale

    def _check_dim(self, Xdim, in_param, paramind=None):
        if Xdim != self.d:
            raise ValueError("Dimensions of data inconsistent!")
        sparam = self.params if paramind is None else self.params[paramind]
        if in_param is None:
            in_param = sparam.value
        sparam.bounds.check(in_param)  # result ignored, as in the reference (:603)
        if np.isscalar(in_param):
            in_param = np.array([in_param])
        if (sparam.shape == () and len(in_param) == 1) or np.shape(in_param) == sparam.shape:
            return in_param
        raise ValueError("Dimension of input parameter is inconsistent!")


# --------------------------------------------------------------------------------------
# Random Fourier feature bases on the GPU (reference: basis_functions.py:818-1208)
# --------------------------------------------------------------------------------------

def _sharded_gram(basis, X, y, params, devices):
    """``basis.gram(X, y, *params, devices=...)``: the statistics of all rows with the row shards resident on the GPUs of
    ``devices`` (one process; revrand_amd/multigpu.py)."""
    from . import multigpu
    if isinstance(basis, BasisCat):
        hyp = list(params) or basis.params_values()
    else:  # one length-scale basis: validated as its own `gram` would (apply_ind slicing happens in device_fit_state)
        hyp = basis._check_dim(basis.d, params[0] if params else None)
    res = multigpu.gram(basis, X, y, hyp, devices)
    if res is None:  # no device-resident route for this basis: its one-GPU statistics
        return basis.gram(X, y, *params)
    return res


_HANDLE_CACHE_LOCK = threading.Lock()


def _handle_cache(basis, name="_hip_handle"):
    """(cache, key) of a basis' device handles: one per process AND per device context (`_hip.device_key`) -- a basis
    whose rows are sharded over the members of a device group (multigpu.ShardedFitState) holds W once on every member.
    Entries of another process (a fork) are dropped; the cache is never pickled."""
    key = _hip.device_key()
    with _HANDLE_CACHE_LOCK:  # (the member threads of a device group build their fit states on the same basis side by side)
        cache = basis.__dict__.get(name)
        if not isinstance(cache, dict) or any(k[0] != key[0] for k in list(cache)):
            cache = basis.__dict__[name] = {}
    return cache, key


class _DevicePosterior(object):
    """Mixin of the fit states: the sufficient statistics [G | b | y^T y] of the last Gram pass and the posterior
    covariance stay in HBM (rr_posterior_dev), so an `_elbo` evaluation moves O(F) numbers over PCIe."""

    def _stats_init(self, dev, F):
        self.dev, self.F = dev, int(F)
        self.acc = dev.malloc((self.F * self.F + self.F + 1) * 8)
        self.dC = self.dCbest = None
        self.best_on_device = False

    def _stat_ptrs(self):
        base, F = self.acc.ptr.value, self.F
        return tuple(_hip.ctypes.c_void_p(base + o * 8) for o in (0, F * F, F * F + F))

    lazy_yty = True   # gram_device(..., want_yty=False): y^T y stays in HBM (a full `_elbo` never reads it: one host round trip less)

    def _finish_stats(self, reduce=None, nrows=0, want_yty=True):
        pG, pb, pt = self._stat_ptrs()
        if reduce is not None:
            # row-sharded fit: pack the upper triangle, ONE ncclAllReduce of [tri G | b | yty | N], unpack into the full
            # symmetric G -- stream-ordered after the Gram kernels, nothing leaves HBM (parallel.RcclComm)
            self.N_total = reduce(self.F, pG, pb, pt, nrows)
        else:
            _hip._check(self.dev.lib, self.dev.lib.rr_symmetrize_dev(self.dev.ctx, pG, self.F))
        if not want_yty:
            return None
        return float(self.dev.download(self.acc, (1,), np.float64, offset_bytes=(self.F * self.F + self.F) * 8)[0])

    def stats_host(self):
        F = self.F
        out = self.dev.download(self.acc, (F * F + F + 1,), np.float64)
        return out[:F * F].reshape(F, F), out[F * F:F * F + F].copy(), float(out[-1])

    def b_host(self):
        """Phi^T y of the last ``gram_device`` (F numbers)."""
        return self.dev.download(self.acc, (self.F,), np.float64, offset_bytes=self.F * self.F * 8)

    def posterior(self, iL, var):
        """(m, diag C, log|iC|, sum(G o C)) from the statistics of the last ``gram_device``; C stays in ``self.dC``.
        None when the Cholesky is not safe (the estimator then takes the host SVD route)."""
        if self.dC is None:
            self.dC = self.dev.malloc(self.F * self.F * 8)
        pG, pb, _ = self._stat_ptrs()
        return self.dev.posterior(self.F, pG, pb, iL, var, self.dC)

    def keep_best(self):
        """The covariance just computed is the best so far: keep it (buffer swap, no copy)."""
        self.dC, self.dCbest = self.dCbest, self.dC
        self.best_on_device = True

    def best_covariance(self):
        return self.dev.download(self.dCbest, (self.F, self.F), np.float64)

    def _stats_release(self):
        for b in (self.acc, self.dC, self.dCbest):
            if b is not None:
                b.free()
        self.acc = self.dC = self.dCbest = None


class DeviceFitState(_DevicePosterior):
    """X and y resident on the GPU for the whole of a ``fit`` (slm.py:118-126 calls ``_elbo``
    ~50-200 times): per evaluation only the (F, F) statistics, the posterior and d+1 scalars cross PCIe.

    ``gram(ls)``            -> (Phi^T Phi, Phi^T y, y^T y)                    slm.py:145-146,157
    ``second_pass(ls, m, C)`` -> (sqErr, dhyp) with dhyp the value ``apply_grad(dhyps, basis.grad(X))``
                               would have (slm.py:161-162,193-197), computed without dPhi.
    """

    def __init__(self, handle, W, X, y, dtype="f32"):
        self.handle, self.W = handle, np.asarray(W, dtype=np.float64)
        # f64 bases keep X, y and every product in float64; float64-phase f32 bases (RandomLaplace) keep X and y in float64
        ft = handle.x_dtype
        self.dX = handle.upload(np.ascontiguousarray(X, dtype=ft))
        self.dy = handle.dev.upload_vector(np.ascontiguousarray(y, dtype=ft))
        self._stats_init(handle.dev, 2 * handle.n)

    def gram(self, lenscale):
        return self.handle.gram_host(self.dX, self.dy, lenscale)

    def gram_device(self, lenscale, reduce=None, want_yty=True):
        """Statistics of this length scale into the resident buffer (summed over ranks by `reduce`); returns
        y^T y (want_yty=False: None, and nothing is waited for)."""
        self.gram_launch(lenscale)
        return self._finish_stats(reduce, self.dX.shape[0], want_yty)

    def gram_launch(self, lenscale):
        """The asynchronous part of ``gram_device``: this state's rows into its (zeroed) accumulators, nothing waited for --
        a device group queues it on every member before it sums the members' statistics (multigpu.ShardedFitState)."""
        self.dev.memset(self.acc)
        pG, pb, pt = self._stat_ptrs()
        self.handle.gram_dev(self.dX, self.dy, lenscale, pG, pb, pt)

    @property
    def nrows(self):
        return self.dX.shape[0]

    def second_pass(self, lenscale, m, C, var):
        sq, T = self.handle.elbo_pass2(self.dX, self.dy, lenscale, m, C)
        ls = np.atleast_1d(np.asarray(lenscale, dtype=float))
        # L-BFGS visits length scales up to the log-space bound (1e+100 and beyond): l^2 or T W may overflow to inf
        # there, exactly as the reference's dPhi products do; the gradient is then 0 or inf, not a warning
        with np.errstate(over="ignore", invalid="ignore"):
            if ls.size == 1:  # the reference's isotropic gradient: input dimension 0 only
                return sq, float((T[0] * self.W[0]).sum() / (var * ls[0] ** 2))
            return sq, (T * self.W).sum(axis=1) / (var * ls ** 2)

    def release(self):
        self.dX.free()
        self.dy.free()
        self._stats_release()


class _ResidentHost(object):
    """Child of a resident concatenated fit whose (narrow, parameter-free) block is made on the host."""

    nparams = 0

    def __init__(self, basis, ncols=0):
        self.basis, self.ncols = basis, ncols

    def put(self, fm, X, r0, rows, col0, params):
        fm.put_host(self.basis.transform(X[r0:r0 + rows]), col0)

    def gather(self, didx, M, dev=None, slot=None):
        pass

    def put_batch(self, fm, M, col0, params, slot=None):
        fm.put_host(self.basis.transform(np.zeros((M, self.ncols))), col0)  # a constant column: only the row count matters

    def release(self):
        pass


class _ResidentLinear(object):
    """LinearBasis child: its X columns stay on the device."""

    nparams = 0

    def __init__(self, basis, X, dtype=None):
        self.onescol = basis.onescol
        self.dX = _hip.get_device().upload_matrix(np.ascontiguousarray(X, dtype=np.float64 if dtype == "f64" else np.float32))

    def put(self, fm, X, r0, rows, col0, params):
        fm.put_linear(_hip.DeviceView(self.dX, r0, rows), self.onescol, col0)

    def gather(self, didx, M, dev=None, slot=None):
        _gather_slot(self, didx, M, dev, slot)

    def put_batch(self, fm, M, col0, params, slot=None):
        fm.put_linear(_hip.DeviceView(_batch_buffer(self, slot), 0, M), self.onescol, col0)

    def release(self):
        self.dX.free()
        _free_batches(self)


class _ResidentRFF(object):
    """Random Fourier (or FastFood, through its dense equivalent) child: X resident in the padded layout,
    T = X^T A accumulated on the device by rr_featmat_pass2_rff."""

    nparams = 1

    def __init__(self, basis, X):
        self.basis = basis
        self.h, W = basis._dense_handle()
        self.W = np.asarray(W, dtype=np.float64)
        self.dX = self.h.upload(np.ascontiguousarray(X, dtype=self.h.x_dtype))
        self.dT = self.h.dev.zeros(self.W.size * 8)

    def put(self, fm, X, r0, rows, col0, params):
        self.ls = self.basis._check_dim(self.basis.d, params[0] if params else None)
        fm.put_rff(self.h, _hip.DeviceView(self.dX, r0, rows), self.ls, col0)

    def gather(self, didx, M, dev=None, slot=None):
        _gather_slot(self, didx, M, dev, slot)

    def put_batch(self, fm, M, col0, params, slot=None):
        self.ls = self.basis._check_dim(self.basis.d, params[0] if params else None)
        fm.put_rff(self.h, _hip.DeviceView(_batch_buffer(self, slot), 0, M), self.ls, col0)

    def batch(self, M):
        """The rows the feature matrix was last filled from: the gathered minibatch, or the first M resident rows."""
        cur = getattr(self, "_cur", None)
        return _hip.DeviceView(cur if cur is not None else self.dX, 0, M)

    def reset(self):
        self.h.dev.memset(self.dT)

    def grad(self, fm, r0, rows, col0):
        fm.pass2_rff(self.h, _hip.DeviceView(self.dX, r0, rows), col0, self.dT)

    def plan(self, fm, r0, rows, col0):
        fm.pass2_plan_rff(self.h, _hip.DeviceView(self.dX, r0, rows), col0, self.dT)

    def dhyp(self, var):
        T = self.h.dev.download(self.dT, self.W.shape, np.float64)
        ls = np.atleast_1d(np.asarray(self.ls, dtype=float))
        # (L-BFGS visits length scales up to the log-space bound: l^2 or T W may overflow there, as in DeviceFitState)
        with np.errstate(over="ignore", invalid="ignore"):
            if ls.size == 1:  # the reference's isotropic gradient: input dimension 0 only
                return float((T[0] * self.W[0]).sum() / (var * ls[0] ** 2))
            return (T * self.W).sum(axis=1) / (var * ls ** 2)

    def release(self):
        self.dX.free()
        self.dT.free()
        _free_batches(self)


class _ResidentFastFood(_ResidentRFF):
    """FastFoodRBF child: Phi comes from the Hadamard / permute / diagonal chain itself (rr_fastfood16_kernel writes it
    straight into the feature matrix, basis_functions.py:1263-1289); the length-scale gradient's contraction T = X^T A
    needs X and Phi only, and meets the dense equivalent W = _makeVX(I_d) on the host in `dhyp` -- the chain is linear
    in x."""

    def __init__(self, basis, X):
        super().__init__(basis, X)
        self.ff = basis._handles()[0]

    def put(self, fm, X, r0, rows, col0, params):
        self.ls = self.basis._check_dim(self.basis.d, params[0] if params else None)
        fm.put_fastfood(self.ff, _hip.DeviceView(self.dX, r0, rows), self.ls, col0)

    def put_batch(self, fm, M, col0, params, slot=None):
        self.ls = self.basis._check_dim(self.basis.d, params[0] if params else None)
        fm.put_fastfood(self.ff, _hip.DeviceView(_batch_buffer(self, slot), 0, M), self.ls, col0)


class _ResidentFastFoodGM(_ResidentRFF):
    """FastFoodGM child (basis_functions.py:1386-1562): the four trig blocks come from the chain kernel's mixture mode
    (rr_featmat_put_fastfood_gm); to the second pass they are two random-Fourier shaped children side by side --
    [cos | sin](VX + mX) at col0 and [cos | sin](VX - mX) at col0 + 2n -- whose contractions T+ / T- = X^T A+- give both
    gradients without the two (N, 4n, d) tensors of :1477-1537:
        sum(E o dPhi/dmean_i) = sum_f (T+ - T-)[i, f],      sum(E o dPhi/dl_i) = -(1 / l_i^2) sum_f V[i, f] (T+ + T-)[i, f]
    with V = _makeVX(I_d) the dense equivalent of the chain (only its d x n numbers, on the host)."""

    nparams = 2

    def __init__(self, basis, X):
        super().__init__(basis, X)  # self.h: the dense handle (n, padded layout of X); self.W = V; self.dT = T+
        self.ff = basis._handles()[0]
        self.dTm = self.h.dev.zeros(self.W.size * 8)
        self.n = self.h.n

    def _params(self, params):
        d = self.basis.d
        self.mean = self.basis._check_dim(d, params[0] if len(params) > 0 else None, paramind=0)
        self.ls = self.basis._check_dim(d, params[1] if len(params) > 1 else None, paramind=1)

    def put(self, fm, X, r0, rows, col0, params):
        self._params(params)
        fm.put_fastfood_gm(self.ff, _hip.DeviceView(self.dX, r0, rows), self.mean, self.ls, col0)

    def put_batch(self, fm, M, col0, params, slot=None):
        self._params(params)
        fm.put_fastfood_gm(self.ff, _hip.DeviceView(_batch_buffer(self, slot), 0, M), self.mean, self.ls, col0)

    def reset(self):
        self.h.dev.memset(self.dT)
        self.h.dev.memset(self.dTm)

    def grad(self, fm, r0, rows, col0):
        view = _hip.DeviceView(self.dX, r0, rows)
        fm.pass2_rff(self.h, view, col0, self.dT)
        fm.pass2_rff(self.h, view, col0 + 2 * self.n, self.dTm)

    def plan(self, fm, r0, rows, col0):
        view = _hip.DeviceView(self.dX, r0, rows)
        fm.pass2_plan_rff(self.h, view, col0, self.dT)
        fm.pass2_plan_rff(self.h, view, col0 + 2 * self.n, self.dTm)

    def glm_grad(self, fm, M, col0):
        """The same two contractions against EdPhi of a GLM minibatch step (glm.py:274-275)."""
        view = self.batch(M)
        fm.glm_rff(self.h, view, col0, self.dT)
        fm.glm_rff(self.h, view, col0 + 2 * self.n, self.dTm)

    def dhyp(self, var):
        """[d/dmean, d/dlenscale] as ``apply_grad(dhyps, basis.grad(X, mean, lenscale))`` would return them."""
        Tp = self.h.dev.download(self.dT, self.W.shape, np.float64)
        Tm = self.h.dev.download(self.dTm, self.W.shape, np.float64)
        ls = np.atleast_1d(np.asarray(self.ls, dtype=float))
        with np.errstate(over="ignore", invalid="ignore"):
            dmean = -(Tp - Tm).sum(axis=1) / var
            dlen = ((Tp + Tm) * self.W).sum(axis=1) / (var * ls ** 2)
        if self.basis.d == 1:  # the reference's d == 1 gradients are 2-d arrays: scalars out of apply_grad
            return [float(dmean[0]), float(dlen[0])]
        return [dmean, dlen]

    def release(self):
        super().release()
        self.dTm.free()


def _gather_batch(dX, dXb, didx, M, dev=None):
    """Rows didx of the resident matrix dX into a (grow-only) batch matrix of the same layout; on `dev`'s stream (default:
    the context the data were uploaded through)."""
    dev = dX.dev if dev is None else dev
    if dXb is None or dXb.shape[0] < M:
        if dXb is not None:
            dXb.free()
        # allocated (and zero-filled) through the context whose stream gathers into it: a fill queued on another stream
        # could land after the gather
        dXb = dev.empty_matrix(M, dX.shape[1], dX.dtype, ld_dev=dX.ld)
    dev.gather_rows(dX, didx, M, dXb)
    return dXb


# A resident child's gathered minibatches: `dXb` (the step gathers for itself) or, when the minibatch worker gathers
# ahead of the step on its own stream, one of a few buffers in turn (`slot`); `_cur` = what the feature matrix was last
# filled from (the gradient contraction of the same step reads it again).
def _gather_slot(child, didx, M, dev, slot):
    if slot is None:
        child.dXb = _gather_batch(child.dX, getattr(child, "dXb", None), didx, M, dev)
    else:
        slots = child.__dict__.setdefault("_slots", {})
        slots[slot] = _gather_batch(child.dX, slots.get(slot), didx, M, dev)


def _batch_buffer(child, slot):
    child._cur = child.dXb if slot is None else child._slots[slot]
    return child._cur


def _free_batches(child):
    for buf in [getattr(child, "dXb", None)] + list(child.__dict__.get("_slots", {}).values()):
        if buf is not None:
            buf.free()
    child.dXb, child._slots, child._cur = None, {}, None


class _ResidentGeneric(object):
    """Any other basis: its block is transformed by its own (GPU or host) ``transform`` and uploaded; its
    gradient contraction is formed on the host from the downloaded EdPhi block."""

    def __init__(self, basis):
        self.basis = basis
        self.nparams = count_args(basis.transform) - 1

    def put(self, fm, X, r0, rows, col0, params):
        self.mine = list(params)
        fm.put_host(self.basis.transform(X[r0:r0 + rows], *params), col0)

    def release(self):
        pass


class MinibatchFeatures(object):
    """Phi of one minibatch (or of query rows) assembled in a device feature matrix for the generalised linear
    model: ``glm_step`` (fs, likelihood derivatives, Edws, EdPhi on the device), ``glm_basis_grads``
    (``apply_grad(lambda dPhi: -(EdPhi * dPhi).sum(), basis.grad(X, *hypers))`` without dPhi for the random
    Fourier family) and ``project`` (latent function samples Phi w)."""

    def __init__(self, basis):
        self.basis = basis
        self.is_cat = isinstance(basis, BasisCat)
        self.bases = basis.bases if self.is_cat else [basis]
        self.fm, self.children, self.dev = None, [], None
        self._stage_bufs, self._targets = {}, None

    def _stage(self, name, arr, dtype):
        """`arr` in this object's grow-only device buffer `name` (no allocation, no free -- hipFree waits for the device --
        per SVI step; the copy is ordered after everything queued on the stream, so the previous step is over)."""
        arr = np.ascontiguousarray(arr, dtype=dtype)
        dev = self.dev if self.dev is not None else _hip.get_device()
        buf = self._stage_bufs.get(name)
        if buf is None or buf.nbytes < arr.nbytes:
            if buf is not None:
                buf.free()
            buf = self._stage_bufs[name] = dev.malloc(max(arr.nbytes, 4))
        if arr.nbytes:
            _hip._check(dev.lib, dev.lib.rr_memcpy_h2d(dev.ctx, buf.ptr, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))
        buf.shape, buf.dtype = arr.shape, arr.dtype
        return buf

    def stage_targets(self, y, rowarg):
        """Upload the step's targets (and per-row likelihood argument) BEFORE its features are launched: the copies
        synchronise the stream, and behind the feature kernels they would make the host wait for them."""
        self._targets = None
        staged = (self._stage("y", y, np.float32), None if rowarg is None else self._stage("rowarg", rowarg, np.float32))
        self._targets = (id(y), len(y), rowarg is None) + staged

    def _take_targets(self, y, rowarg):
        """The targets staged for THIS step's y (same array, same length, same kind of row argument); anything else --
        targets left behind by a step that failed between staging and its kernels -- is uploaded afresh."""
        t, self._targets = self._targets, None
        if t is not None and t[:3] == (id(y), len(y), rowarg is None):
            return t[3], t[4]
        return self._stage("y", y, np.float32), None if rowarg is None else self._stage("rowarg", rowarg, np.float32)

    def _ensure(self, rows, F):
        if self.fm is None or self.fm.max_rows < rows or self.fm.F != F:
            self.fm = None  # free the old one first
            self.fm = _hip.FeatureMatrix(rows, F)
            self.dev = self.fm.dev

    def _drop_children(self):
        if not getattr(self, "resident", False):
            for c, _, _ in self.children:
                c.release()
        self.children = []

    def make_resident(self, X):
        """Keep every child's columns of X on the device for a whole fit (minibatches are then gathered there by
        index); False -- and nothing kept -- if a child cannot."""
        self._drop_children()
        kids = []
        for b in self.bases:
            c = b._resident_child(X)
            if c is None:
                for k in kids:
                    k.release()
                return False
            kids.append(c)
        self._kids = kids
        self._dims = [int(b.get_dim(X)) for b in self.bases]
        self.resident = True
        return True

    def assemble_idx(self, idx, hypers, gathered=None):
        """`assemble` for rows `idx` of the resident data; `gathered`: the token of `prefetch_batch` for these rows (the
        index upload and the gathers are done already, in buffer set `gathered.slot`)."""
        self.children = []
        M = len(idx)
        self._ensure(M, int(sum(self._dims)))
        slot = None if gathered is None else gathered.slot
        didx = self._stage("idx", idx, np.int32) if gathered is None else None
        self.fm.begin(M)
        args, col0 = list(hypers), 0
        for child, w in zip(self._kids, self._dims):
            mine, args = args[:child.nparams], args[child.nparams:]
            if gathered is None:
                child.gather(didx, M)
            child.put_batch(self.fm, M, col0, mine, slot=slot)
            self.children.append((child, col0, w))
            col0 += w
        self.M = M  # (no synchronisation: the gathers and feature kernels run while the host prepares the step)

    def batch_rows(self, idx, gathered=None):
        """Every resident child's rows `idx` of its data on the device, in concatenation order (the resident SVI loop,
        glm._ResidentLoop): gathered ahead by `prefetch_batch` (its slot) or now."""
        M = len(idx)
        if gathered is None:
            didx = self._stage("idx", idx, np.int32)
            for child in self._kids:
                child.gather(didx, M)
        return [_hip.DeviceView(_batch_buffer(child, None if gathered is None else gathered.slot), 0, M) for child in self._kids]

    PREFETCH_SLOTS = 3  # the worker runs up to two steps ahead of the step on the device (6 under the resident loop, whose
    #                     host queues steps up to two ahead of the device on top of that)

    def prefetch_batch(self, updev, idx, y, rowarg):
        """On the minibatch worker thread, for a FUTURE step: upload the row indices, gather every child's rows and upload
        the targets -- on the upload context's stream (`updev`, a second context of the same GPU), into one of
        PREFETCH_SLOTS buffer sets in turn, finished before this returns.  The step then starts with its feature kernels:
        none of these copies (each synchronises a stream) is left on its critical path."""
        slot = self.__dict__.get("_pf_turn", 0) % self.PREFETCH_SLOTS
        self._pf_turn = self.__dict__.get("_pf_turn", 0) + 1
        M = len(idx)

        def stage(name, arr, dtype):
            arr = np.ascontiguousarray(arr, dtype=dtype)
            key = (name, slot)
            buf = self._stage_bufs.get(key)
            if buf is None or buf.nbytes < arr.nbytes:
                if buf is not None:
                    buf.free()
                buf = self._stage_bufs[key] = updev.malloc(max(arr.nbytes, 4))
            if arr.nbytes:
                _hip._check(updev.lib, updev.lib.rr_memcpy_h2d(updev.ctx, buf.ptr, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))
            buf.shape, buf.dtype = arr.shape, arr.dtype
            return buf
        didx = stage("idx", idx, np.int32)
        for child in self._kids:
            child.gather(didx, M, dev=updev, slot=slot)
        dy = stage("y", y, np.float32)
        dn = None if rowarg is None else stage("rowarg", rowarg, np.float32)
        updev.sync()
        return _Gathered(slot, M, dy, dn, id(y), rowarg is None)

    def take_prefetched_targets(self, gathered, y, rowarg):
        """The step's targets are the ones `prefetch_batch` uploaded (checked: same array, same kind of row argument)."""
        if gathered.key != (id(y), len(y), rowarg is None):
            return False  # not this step's arrays (a wrapper copied them): the step uploads its own
        self._targets = gathered.key + (gathered.dy, gathered.dn)
        return True

    def assemble(self, X, hypers):
        self._targets = None  # this route never stages ahead
        self._drop_children()
        M = X.shape[0]
        dims = [int(b.get_dim(X)) for b in self.bases]
        self._ensure(M, int(sum(dims)))
        self.fm.begin(M)
        args, col0 = list(hypers), 0
        for b, w in zip(self.bases, dims):
            child = b._resident_child(X)
            if child is None:
                child = _ResidentGeneric(b)
            mine, args = args[:child.nparams], args[child.nparams:]
            child.put(self.fm, X, 0, M, col0, mine)
            self.children.append((child, col0, w))
            col0 += w
        self.M = M

    def _plan_basis_grads(self, objective_only=False):
        """A lone random Fourier child: tell the step which contraction `glm_basis_grads` will ask for, so that it can
        form it from the blocks of EdPhi while they are in registers (rr_featmat_glm_plan_rff); dT is zeroed here."""
        self._planned = False
        if objective_only or len(self.children) != 1:
            return
        child, col0, _ = self.children[0]
        if isinstance(child, _ResidentRFF) and not isinstance(child, _ResidentFastFoodGM):
            child.reset()
            self.fm.glm_plan_rff(child.h, child.batch(self.M), col0, child.dT)
            self._planned = True

    def glm_step(self, y, rowarg, lik, lik_param, WS, K, L):
        dy, dn = self._take_targets(y, rowarg)
        self._plan_basis_grads()
        return self.fm.glm_step(dy, dn, lik, lik_param, WS, K, L)

    supports_objective_only = True  # glm_step_sampled / glm_step_draws take objective_only=True (no gradient GEMMs)
    accepts_device_draws = True     # glm_step_draws takes E as a float32 (K L, F) DeviceBuffer as well

    def glm_step_sampled(self, y, rowarg, lik, lik_param, m, C, K, L, seed, step, objective_only=False):
        dy, dn = self._take_targets(y, rowarg)
        self._plan_basis_grads(objective_only)
        return self.fm.glm_step_sampled(dy, dn, lik, lik_param, m, C, K, L, seed, step, objective_only)

    def glm_step_draws(self, y, rowarg, lik, lik_param, m, C, K, L, E, objective_only=False):
        dy, dn = self._take_targets(y, rowarg)
        self._plan_basis_grads(objective_only)
        return self.fm.glm_step_draws(dy, dn, lik, lik_param, m, C, K, L, E, objective_only)

    def glm_basis_grads(self, X):
        grads = []
        for child, col0, w in self.children:
            if isinstance(child, _ResidentFastFoodGM):
                child.reset()
                child.glm_grad(self.fm, self.M, col0)
                g = child.dhyp(1.0)
            elif isinstance(child, _ResidentRFF):
                if not self.__dict__.pop("_planned", False):  # (planned: dT was zeroed before the step, which may have
                    child.reset()                              # accumulated it already -- glm_rff then returns at once)
                self.fm.glm_rff(child.h, child.batch(self.M), col0, child.dT)
                g = [child.dhyp(1.0)]   # -(E o dPhi_i).sum() = +(1/l_i^2) W[i,:].T[i,:]
            elif child.nparams:
                E = self.fm.glm_edphi(self.M, col0, w)
                g = apply_grad(lambda dPhi: -(E * dPhi).sum(), child.basis.grad(X, *child.mine))
                if not self.is_cat:
                    return g
                g = atleast_list(g)
            else:
                g = []
            grads.extend(g)
        if not self.is_cat:   # (a lone spectral-mixture component has TWO gradients: [dmean, dlenscale])
            return (grads[0] if len(grads) == 1 else grads) if grads else []
        return grads if len(grads) != 1 else grads[0]

    def project(self, X, hypers, W):
        """Phi(X) W, (N, S), in row chunks."""
        N = X.shape[0]
        F = int(sum(int(b.get_dim(X)) for b in self.bases))
        Fp = (F + 255) // 256 * 256
        chunk = int(max(256, min(N, (8 << 30) // (8 * Fp))))
        out = np.empty((N, W.shape[1]))
        for r0 in range(0, N, chunk):
            Xc = X[r0:r0 + chunk]
            self.assemble(Xc, hypers)
            out[r0:r0 + chunk] = self.fm.project(Xc.shape[0], W)
        return out

    def release(self):
        if getattr(self, "resident", False):
            for k in self._kids:
                k.release()
            self.resident = False
        self._drop_children()
        for buf in self._stage_bufs.values():
            buf.free()
        self._stage_bufs, self._targets = {}, None
        self.fm = None


class _Gathered(object):
    """A minibatch made ready on the device ahead of its step (MinibatchFeatures.prefetch_batch)."""

    def __init__(self, slot, M, dy, dn, yid, no_rowarg):
        self.slot, self.M, self.dy, self.dn = slot, M, dy, dn
        self.key = (yid, M, no_rowarg)


class CatFitState(_DevicePosterior):
    """DeviceFitState for a BasisCat: every child keeps its columns of X on the GPU, Phi is assembled in a
    device feature matrix per row chunk; same ``gram`` / ``second_pass`` / ``release`` interface, ``dhyp``
    structured like ``apply_grad(f, cat.grad(X, *hypers))``."""

    def __init__(self, cat, children, X, y, chunk_rows=None, dtype="f32"):
        self.X, self.N = X, X.shape[0]
        self.children = children
        self.dtype = dtype  # "f64": float64 feature matrix, statistics and second pass (rr_featmat64_*)
        self.F = int(cat.get_dim(X))
        self.ends = [int(e) for e in np.cumsum([0] + [int(b.get_dim(X)) for b in cat.bases])]
        self.dev = _hip.get_device()
        es = 8 if dtype == "f64" else 4
        self.dy = self.dev.upload_vector(np.ascontiguousarray(y, dtype=np.float64 if dtype == "f64" else np.float32))
        Fp = (self.F + 255) // 256 * 256
        if chunk_rows is None:  # P, its transpose and U = P C: 3 Fp elements per row, within ~24 GiB
            chunk_rows = (24 << 30) // (3 * es * Fp)
        self.chunk = int(max(256, min(self.N, chunk_rows)))
        if self.N > self.chunk:
            # several chunks: equal ones (config 3's 1.25M-row shard: 5 x 250 000 instead of 4 x 254 200 + 233 200), like the
            # single-basis passes (rr_rff.hip gram_run, rr_elbo.hip pass2_run)
            nchunks = -(-self.N // self.chunk)
            self.chunk = -(-self.N // nchunks)
        self.fm = (_hip.FeatureMatrix64 if dtype == "f64" else _hip.FeatureMatrix)(self.chunk, self.F)
        self._filled = None
        self._stats_init(self.dev, self.F)

    def _fill(self, r0, rows, hypers):
        key = (r0, rows, tuple(np.asarray(h, dtype=float).tobytes() for h in hypers))
        if self._filled == key:  # single-chunk fits: the second pass reuses the Gram pass' features
            return
        self.fm.begin(rows)
        args = list(hypers)
        for child, col0 in zip(self.children, self.ends):
            mine, args = args[:child.nparams], args[child.nparams:]
            child.put(self.fm, self.X, r0, rows, col0, mine)
        self._filled = key

    def _chunks(self):
        for r0 in range(0, self.N, self.chunk):
            yield r0, min(self.chunk, self.N - r0)

    def gram_device(self, hypers, reduce=None, want_yty=True):
        """Statistics of these hyper-parameters into the resident buffer (summed over ranks by `reduce`);
        returns y^T y (want_yty=False: None, and nothing is waited for)."""
        self.gram_launch(hypers)
        return self._finish_stats(reduce, self.N, want_yty)

    def gram_launch(self, hypers):
        """The part of ``gram_device`` before the statistics are summed / mirrored (see DeviceFitState.gram_launch)."""
        hypers = atleast_list(hypers)
        self.dev.memset(self.acc)
        pG, pb, pt = self._stat_ptrs()
        for r0, rows in self._chunks():
            self._fill(r0, rows, hypers)
            self.fm.gram_into(_hip.DeviceView(self.dy, r0, rows), pG, pb, pt)

    @property
    def nrows(self):
        return self.N

    def gram(self, hypers):
        self.gram_device(hypers)
        return self.stats_host()

    def second_pass(self, hypers, m, C, var):
        hypers = atleast_list(hypers)
        self.fm.pass2_begin(m, C)
        with_grad = [(c, col0) for c, col0 in zip(self.children, self.ends) if c.nparams]
        for c, _ in with_grad:
            c.reset()
        # every consumer of U = Phi C known ahead (all of them random Fourier children): the product may contract itself
        # with each child's block in registers instead of being stored (rr_featmat_pass2_rows_planned)
        planned = bool(with_grad) and hasattr(self.fm, "pass2_rows_planned") and \
            all(isinstance(c, _ResidentRFF) for c, _ in with_grad)
        for r0, rows in self._chunks():
            self._fill(r0, rows, hypers)
            if planned:
                for c, col0 in with_grad:
                    c.plan(self.fm, r0, rows, col0)
                self.fm.pass2_rows_planned(_hip.DeviceView(self.dy, r0, rows))
            else:
                self.fm.pass2_rows(_hip.DeviceView(self.dy, r0, rows))
            for c, col0 in with_grad:
                c.grad(self.fm, r0, rows, col0)
        sq = self.fm.pass2_end()
        grads = []
        for c, _ in with_grad:  # in concatenation order, one entry per parameter (a mixture component has two)
            g = c.dhyp(var)
            grads.extend(g) if isinstance(g, list) else grads.append(g)
        return sq, (grads if len(grads) != 1 else grads[0])

    def release(self):
        for c in self.children:
            c.release()
        self.dy.free()
        self.fm = None
        self._stats_release()


class _RandomKernelBasis(_LengthScaleBasis):
    """Phi = [cos(X W/l), sin(X W/l)]/sqrt(nbases); subclasses only sample W."""

    _default_dtype = "f32"
    _heavy_tailed = False  # W drawn from a distribution without a variance (RandomLaplace): "f32" means "f32p64"

    @slice_init
    def __init__(self, nbases, Xdim, lenscale=Parameter(gamma(1.), Positive()), regularizer=None,
                 random_state=None, dtype=None):
        dtype = self._default_dtype if dtype is None else dtype
        if dtype not in ("f32", "f64", "f32p64"):
            raise ValueError("dtype must be 'f32', 'f64' or 'f32p64'")
        self.d = Xdim
        self.n = nbases
        # "f32p64": the f32 pipeline (features, Gram, second pass, feature matrix, GLM step all float32) with the phase
        # x . w / l accumulated and reduced in float64 (RR_F32P64).  `dtype` stays "f32" -- every f32 route takes the basis.
        self.phase64 = dtype == "f32p64" or (dtype == "f32" and self._heavy_tailed and Xdim <= 128)
        if self.phase64 and Xdim > 128:
            raise ValueError("dtype 'f32p64' needs Xdim <= 128 (use 'f64')")
        dtype = "f64" if (dtype == "f32" and self._heavy_tailed and not self.phase64) else dtype
        self.dtype = "f32" if dtype == "f32p64" else dtype
        self.random_state = random_state  # for repr
        self._random = check_random_state(random_state)
        self.W = self._weightsamples()
        self._init_lenscale(lenscale)
        super(_LengthScaleBasis, self).__init__(regularizer)

    # device handle: created on first use in this process, never pickled
    def _handle(self):
        cache, key = _handle_cache(self)
        h = cache.get(key)
        if h is None:
            compute = "f32p64" if getattr(self, "phase64", False) else self.dtype
            h = cache[key] = _hip.RffHandle(self.W, compute=compute)
        return h

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_hip_handle", None)
        return state

    def get_dim(self, X):
        return 2 * self.n

    @slice_transform
    def transform(self, X, lenscale=None):
        """(N, 2*nbases) float64, cos block then sin block (basis_functions.py:838-864)."""
        N, D = X.shape
        lenscale = self._check_dim(D, lenscale)
        return self._handle().transform(X, lenscale)

    @slice_transform
    def grad(self, X, lenscale=None):
        """dPhi/dl: (N, 2*nbases) for a scalar length scale -- dimension 0's contribution only,
        exactly as the reference computes it -- or (N, 2*nbases, d) for ARD
        (basis_functions.py:866-901)."""
        N, D = X.shape
        lenscale = self._check_dim(D, lenscale)
        return self._handle().grad(X, lenscale)

    def gram(self, X, y=None, lenscale=None, devices=None):
        """Fused (Phi^T Phi, Phi^T y, y^T y) without materialising Phi (slm.py:145-146,157).  devices: row shards on
        several GPUs of this process, summed in HBM (multigpu.gram; same result to float64 rounding)."""
        if devices is not None:
            return _sharded_gram(self, X, y, [lenscale], devices)
        return self._gram_one(X, y, lenscale)

    @slice_transform
    def _gram_one(self, X, y=None, lenscale=None):
        N, D = X.shape
        lenscale = self._check_dim(D, lenscale)
        return self._handle().gram(X, y, lenscale)

    @slice_transform
    def _put_features(self, X, fm, col0, lenscale=None):
        lenscale = self._check_dim(X.shape[1], lenscale)
        h = self._handle()
        dX = h.upload(np.ascontiguousarray(X, dtype=h.x_dtype))
        fm.put_rff(h, dX, lenscale, col0)
        fm.dev.sync()
        dX.free()

    @slice_transform
    def grad_contract(self, X, E, lenscale=None):
        """``apply_grad(lambda dPhi: (E * dPhi).sum(), self.grad(X, lenscale))`` without the gradient
        tensor (glm.py:274-275, slm.py:193-197): a scalar for an isotropic length scale (input dimension 0
        only, as the reference computes it), else a (d,) array.  f32 arithmetic on the GPU."""
        lenscale = self._check_dim(X.shape[1], lenscale)
        T = self._handle().grad_contract(X, E, lenscale)
        ls = np.asarray(lenscale, dtype=float)
        if ls.size == 1:
            return float(-(T[0] * self.W[0]).sum() / ls[0] ** 2)
        return -(T * self.W).sum(axis=1) / ls ** 2

    @slice_transform
    def device_fit_state(self, X, y):
        """Upload (X, y) once for a fit (float32, or float64 for dtype="f64" bases)."""
        if X.shape[1] != self.d:
            return None
        return DeviceFitState(self._handle(), self.W, X, y, dtype=self.dtype)

    def _dense_handle(self):
        """(RffHandle, W) whose kernels produce this basis' features."""
        return self._handle(), self.W

    @slice_transform
    def _resident_child(self, X, dtype=None):
        # float64 bases join a float64 state only; f32 bases join either (a float64 feature matrix evaluates them in float64)
        if (self.dtype != "f32" and dtype != "f64") or X.shape[1] != self.d or (dtype == "f64" and self.d > 128):
            return None
        return _ResidentRFF(self, X)

    _predict_checks_rows = True  # `predict_moments` runs the caller's row validation itself (chunk by chunk, under the GPU's work)

    @slice_transform
    def predict_moments(self, X, lenscale, m, C, check_rows=None):
        """(Phi m, rowsum((Phi C) o Phi)) on the device (slm.py:240-243), in the basis' arithmetic.  `check_rows`: the
        estimator's validation of the query rows (`check_array`), applied here -- to the row chunks of a large query as they
        are uploaded, see RffHandle.predict.  C = None asks for the mean alone: (Phi m, None), or None when the basis'
        arithmetic has no such route (float64, float64 phases, Xdim > 128)."""
        lenscale = self._check_dim(X.shape[1], lenscale)
        h = self._dense_handle()[0]
        if C is None and not h.mean_only_ok:
            return None
        return h.predict(X, lenscale, m, C, check_rows=check_rows)

    def __repr__(self):
        return "{}(nbases={}, Xdim={}, lenscale={}, regularizer={}, random_state={})".format(
            type(self).__name__, self.n, self.d, self.params, self.regularizer, self.random_state)


class RandomRBF(_RandomKernelBasis):
    """RBF kernel features: W ~ N(0, 1) (basis_functions.py:916-954)."""

    def _weightsamples(self):
        return self._random.randn(self.d, self.n)


class RandomLaplace(_RandomKernelBasis):
    """Laplace kernel features: W ~ Cauchy (basis_functions.py:957-995).

    Cauchy draws reach |W| ~ 1e5 at nbases = 2048, i.e. phases of ~1e5 revolutions, which an f32
    accumulator resolves to only ~1e-2 rad (measured: 6e-2 normwise error of Phi at d=32, nbases=2048).  For this
    class ``dtype="f32"`` (the default) therefore selects the float64-PHASE variant of the f32 pipeline (``"f32p64"``,
    RR_F32P64: x . w / l accumulated on the f64 matrix cores and reduced modulo one revolution in float64, then float32
    sin / cos): 2e-7 normwise error of Phi, and the basis takes every resident f32 route (fused Gram, concatenated fit
    states, GLM step).  Resident X is kept in float64 for it.  ``dtype="f64"`` is the reference's arithmetic end to end.
    """
    _heavy_tailed = True

    def _weightsamples(self):
        return self._random.standard_cauchy(size=(self.d, self.n))


class RandomCauchy(_RandomKernelBasis):
    """Cauchy kernel features: Gaussian scaled per frequency by sqrt(2 Gamma(1))
    (basis_functions.py:998-1045)."""

    def _weightsamples(self):
        gauss = self._random.randn(self.d, self.n)
        mix = self._random.standard_gamma(1., size=(1, self.n))
        return gauss * np.sqrt(2 * mix)


class _RandomMatern(_RandomKernelBasis):
    """Multivariate-t frequencies with 2p+1 degrees of freedom (basis_functions.py:1048-1065)."""

    def _maternweight(self, p):
        df = 2 * (p + 0.5)
        gauss = self._random.randn(self.d, self.n)
        chi2 = self._random.chisquare(df, size=(self.n,))
        return gauss * np.sqrt(df / chi2)


class RandomMatern32(_RandomMatern):
    """Matern 3/2 features (basis_functions.py:1068-1107)."""

    def _weightsamples(self):
        return self._maternweight(p=1)


class RandomMatern52(_RandomMatern):
    """Matern 5/2 features (basis_functions.py:1110-1150)."""

    def _weightsamples(self):
        return self._maternweight(p=2)


class OrthogonalRBF(_RandomKernelBasis):
    """Orthogonal random features (basis_functions.py:1153-1208): QR blocks, chi row scales."""

    def _weightsamples(self):
        reps = int(np.ceil(self.n / self.d))
        Q = np.empty((self.d, self.d * reps))
        for r in range(reps):
            Q[:, r * self.d:(r + 1) * self.d] = qr(self._random.randn(self.d, self.d))[0]
        S = np.sqrt(self._random.chisquare(df=self.d, size=self.d))
        return S[:, np.newaxis] * Q[:, :self.n]


# --------------------------------------------------------------------------------------
# FastFood (reference: basis_functions.py:1211-1383)
# --------------------------------------------------------------------------------------

class FastFoodRBF(_LengthScaleBasis):
    """FastFood approximation of the RBF kernel: V = S H G PI H B in k blocks of d2 = 2^ceil(log2 d).

    ``transform`` runs the Hadamard / permute / diagonal chain on the GPU (``rr_fastfood_transform``).
    The chain is linear in x, so ``grad`` and ``gram`` go through the random-Fourier kernels with the
    dense equivalent ``W = _makeVX(I_d)`` (itself produced by the chain on the device).
    """

    @slice_init
    def __init__(self, nbases, Xdim, lenscale=Parameter(gamma(1.), Positive()), regularizer=None,
                 random_state=None, dtype="f32"):
        if dtype not in ("f32", "f64"):
            raise ValueError("dtype must be 'f32' or 'f64'")
        self.dtype = dtype
        self.random_state = random_state  # for repr
        self._random = check_random_state(random_state)
        self._init_dims(nbases, Xdim)
        self._init_lenscale(lenscale)
        self._init_matrices()
        super(_LengthScaleBasis, self).__init__(regularizer)

    def _init_dims(self, nbases, Xdim):
        l = int(np.ceil(np.log2(Xdim)))
        self.nbases = nbases
        self.d = Xdim
        self.d2 = pow(2, l)
        self.k = int(np.ceil(nbases / self.d2))
        self.n = self.d2 * self.k

    def _init_matrices(self):
        # draw order B -> G -> PI -> S (basis_functions.py:1346-1350)
        shape = (self.k, self.d2)
        self.B = self._random.randint(2, size=shape) * 2 - 1
        self.G = self._random.randn(*shape)
        self.PI = np.array([self._random.permutation(self.d2) for _ in range(self.k)])
        self.S = self._weightsamples()

    def _weightsamples(self):
        s = np.sqrt(self._random.chisquare(self.d2, size=self.G.shape))
        return self.d2 * s / np.sqrt((self.G ** 2).sum(axis=1))[:, np.newaxis]

    class _LazyDense(object):
        """The dense-equivalent random Fourier handle, built on first use (the chain kernel alone serves `transform`
        and `_makeVX`; `grad`, the Gram and resident fits use this handle)."""

        def __init__(self, owner, ff):
            self.owner, self.ff, self.rff, self.V = owner, ff, None, None

        def get(self):
            if self.rff is None:
                self.V = self.ff.vx(np.eye(self.owner.d), 1.0)  # (d, n): dense equivalent of the chain
                self.rff = _hip.RffHandle(self.V, compute=self.owner.dtype, device=self.ff.dev)
            return self.rff

        def __getattr__(self, name):  # h.grad(...), h.gram(...), h.upload(...), ... on the dense handle
            return getattr(self.get(), name)

    def _handles(self):
        cache, key = _handle_cache(self)
        h = cache.get(key)
        if h is None:
            ff = _hip.FastFoodHandle(self.d, self.d2, self.k, self.B, self.G, self.PI, self.S, compute=self.dtype)
            h = cache[key] = (ff, FastFoodRBF._LazyDense(self, ff))
        return h

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_hip_handle", None)
        return state

    def get_dim(self, X):
        return 2 * self.n

    def _makeVX(self, X):
        """(N, n) structured projection of X (basis_functions.py:1356-1371)."""
        return self._handles()[0].vx(X, 1.0)

    @slice_transform
    def transform(self, X, lenscale=None):
        """(N, 2*n) float64, n = d2*k >= nbases (basis_functions.py:1263-1289)."""
        lenscale = self._check_dim(X.shape[1], lenscale)
        return self._handles()[0].transform(X, lenscale)

    @slice_transform
    def grad(self, X, lenscale=None):
        """dPhi/dl, (N, 2n) or (N, 2n, d); same isotropic quirk as the reference (:1291-1329)."""
        lenscale = self._check_dim(X.shape[1], lenscale)
        return self._handles()[1].grad(X, lenscale)

    def gram(self, X, y=None, lenscale=None, devices=None):
        if devices is not None:
            return _sharded_gram(self, X, y, [lenscale], devices)
        return self._gram_one(X, y, lenscale)

    @slice_transform
    def _gram_one(self, X, y=None, lenscale=None):
        lenscale = self._check_dim(X.shape[1], lenscale)
        return self._handles()[1].gram(X, y, lenscale)

    @slice_transform
    def _put_features(self, X, fm, col0, lenscale=None):
        lenscale = self._check_dim(X.shape[1], lenscale)
        h = self._handles()[1]
        dX = h.upload(np.ascontiguousarray(X, dtype=np.float32))
        fm.put_rff(h, dX, lenscale, col0)
        fm.dev.sync()
        dX.free()

    def _dense_handle(self):
        lazy = self._handles()[1]
        return lazy.get(), lazy.V

    @slice_transform
    def _resident_child(self, X, dtype=None):
        """This basis' share of a device-resident fit (f32 states): features by the chain kernel."""
        if self.dtype != "f32" or dtype == "f64" or X.shape[1] != self.d or type(self)._resident_via_chain is False:
            return None
        return _ResidentFastFood(self, X)

    _resident_via_chain = True

    @slice_transform
    def device_fit_state(self, X, y):
        """(X, y) resident for a whole fit.  f32: the statistics pass runs the CHAIN (rr_fastfood16_kernel -> feature matrix
        -> MFMA SYRK) and the second pass contracts X^T A against the same features -- a one-child CatFitState.  f64 (or
        RR_FASTFOOD_FIT=dense, A/B runs): the dense equivalent W through the random Fourier kernels."""
        if X.shape[1] != self.d:
            return None
        if self.dtype == "f32" and self._resident_via_chain and _hip.os.environ.get("RR_FASTFOOD_FIT", "chain") != "dense":
            import types
            return CatFitState(types.SimpleNamespace(get_dim=self.get_dim, bases=[self]), [_ResidentFastFood(self, X)], X, y)
        return DeviceFitState(*self._dense_handle(), X=X, y=y, dtype=self.dtype)

    def __repr__(self):
        return "{}(nbases={}, Xdim={}, lenscale={}, regularizer={}, random_state={})".format(
            type(self).__name__, self.nbases, self.d, self.params, self.regularizer, self.random_state)


class FastFoodGM(FastFoodRBF):
    """One Gaussian spectral-mixture component ("A la Carte"), basis_functions.py:1386-1562:
    ``Phi = [cos(VX + mX), sin(VX + mX), cos(VX - mX), sin(VX - mX)] / sqrt(2n)`` with ``mX = X.mean``.
    Parameters are always (d,)-shaped ``mean`` and ``lenscale`` (scalars are broadcast, :1539-1549);
    ``grad`` returns the pair (dPhi/dmean, dPhi/dlenscale).  Runs on the dense equivalent of the
    FastFood chain (``rr_gm_transform`` / ``rr_gm_grad``).
    """

    @slice_init
    def __init__(self, nbases, Xdim, mean=Parameter(norm_dist(), Bound()), lenscale=Parameter(gamma(1.), Positive()),
                 regularizer=None, random_state=None, dtype="f32"):
        if dtype not in ("f32", "f64"):
            raise ValueError("dtype must be 'f32' or 'f64'")
        self.dtype = dtype
        self.random_state = random_state  # for repr
        self._random = check_random_state(random_state)
        self._init_dims(nbases, Xdim)
        self._params = [self._init_param(mean), self._init_param(lenscale)]
        self._init_matrices()
        super(_LengthScaleBasis, self).__init__(regularizer)

    def _init_param(self, param):
        if param.shape == (self.d,):
            return param
        if param.shape in ((), (1,)):
            # broadcast to (d,).  The reference mutates `param` in place (:1543-1546), which corrupts the
            # shared default argument for the next basis of a different Xdim; build a new Parameter instead.
            if param.is_random:
                return Parameter(param.dist, param.bounds, shape=(self.d,))
            return Parameter(np.ones(self.d) * param.value, param.bounds)
        raise ValueError("Parameter dimension doesn't agree with X dimensions!")

    def get_dim(self, X):
        return 4 * self.n

    @slice_transform
    def transform(self, X, mean=None, lenscale=None):
        """(N, 4*n) float64 (basis_functions.py:1443-1475): the chain kernel's mixture mode (VX by the Hadamard / permute /
        diagonal chain, mX = X . mean next to it in registers); block sizes it does not serve (d2 < 16, i.e. Xdim <= 8) go
        through the dense equivalent of the chain."""
        mean = self._check_dim(X.shape[1], mean, paramind=0)
        lenscale = self._check_dim(X.shape[1], lenscale, paramind=1)
        ff, dense = self._handles()
        if ff.gm_chain_ok:
            return ff.gm_transform(X, mean, lenscale)
        return dense.gm_transform(X, mean, lenscale)

    @slice_transform
    def grad(self, X, mean=None, lenscale=None):
        """(dPhi/dmean, dPhi/dlenscale), each (N, 4n, d) -- (N, 4n) when d == 1 (:1477-1537)."""
        mean = self._check_dim(X.shape[1], mean, paramind=0)
        lenscale = self._check_dim(X.shape[1], lenscale, paramind=1)
        return self._handles()[1].gm_grad(X, mean, lenscale)

    # -- the fused statistics / resident paths (round 5): the four blocks written by the chain kernel into the device
    #    feature matrix, both gradients contracted on the device (_ResidentFastFoodGM) ------------------------------
    def _chain_fit_ok(self):
        return self.dtype == "f32" and self._handles()[0].gm_chain_ok

    @slice_transform
    def _resident_child(self, X, dtype=None):
        if not self._chain_fit_ok() or dtype == "f64" or X.shape[1] != self.d:
            return None
        return _ResidentFastFoodGM(self, X)

    @slice_transform
    def device_fit_state(self, X, y):
        """(X, y) resident for a whole fit: a one-child CatFitState whose child writes the four blocks with the chain kernel."""
        if not self._chain_fit_ok() or X.shape[1] != self.d:
            return None
        import types
        return CatFitState(types.SimpleNamespace(get_dim=self.get_dim, bases=[self]), [_ResidentFastFoodGM(self, X)], X, y)

    @slice_transform
    def _put_features(self, X, fm, col0, mean=None, lenscale=None):
        mean = self._check_dim(X.shape[1], mean, paramind=0)
        lenscale = self._check_dim(X.shape[1], lenscale, paramind=1)
        ff, dense = self._handles()
        if not ff.gm_chain_ok or self.dtype != "f32":
            return fm.put_host(dense.gm_transform(X, mean, lenscale), col0)
        dX = fm.dev.upload_matrix(np.ascontiguousarray(X, dtype=np.float32))
        fm.put_fastfood_gm(ff, dX, mean, lenscale, col0)
        fm.dev.sync()
        dX.free()

    def gram(self, X, y=None, mean=None, lenscale=None, devices=None):
        """(Phi^T Phi, Phi^T y, y^T y) with Phi assembled on the device by the chain kernel (slm.py:145-146,157); None when
        this basis' arithmetic has no such route (float64, d2 < 16): the estimator then takes the dense Gram of `transform`."""
        if self.dtype != "f32":
            return None
        return BasisCat._of([self]).gram(X, y, mean, lenscale, devices=devices)

    def predict_moments(self, X, hypers, m, C):
        """(Phi m, rowsum((Phi C) o Phi)) with Phi assembled on the device (slm.py:240-243); hypers = [mean, lenscale]."""
        if self.dtype != "f32":
            return None
        return BasisCat._of([self]).predict_moments(X, hypers, m, C)

    def __repr__(self):
        return "{}(nbases={}, Xdim={}, mean={}, lenscale={}, regularizer={}, random_state={})".format(
            type(self).__name__, self.nbases, self.d, self.params[0], self.params[1], self.regularizer,
            self.random_state)


# --------------------------------------------------------------------------------------
# Concatenation (reference: basis_functions.py:1569-1790)
# --------------------------------------------------------------------------------------

class BasisCat(object):
    """Column-wise concatenation of bases; parameters are routed positionally in
    concatenation order, gradients are zero-padded to full width and yielded lazily."""

    def __init__(self, basis_list):
        def merge(blist, b):
            rlist = atleast_list(blist)
            if isinstance(b, BasisCat):
                rlist.extend(b.bases)
            else:
                rlist.append(b)
            return rlist
        self.bases = reduce(merge, basis_list)
        self.__dims = None
        self.__baseinds = None
        self.__slices = None

    @classmethod
    def _of(cls, bases):
        """A concatenation of exactly these bases (the constructor's `reduce` needs two to make a list)."""
        cat = cls.__new__(cls)
        cat.bases = list(bases)
        cat.__dims = cat.__baseinds = cat.__slices = None
        return cat

    def transform(self, X, *params):
        Phi, args = [], list(params)
        for base in self.bases:
            phi, args = base._transform_popargs(X, *args)
            Phi.append(phi)
        return np.hstack(Phi)

    def grad(self, X, *params):
        N = X.shape[0]
        D = self.get_dim(X)
        ends = self.__base_locations(X)
        args = list(params)
        for i, base in enumerate(self.bases):
            g, args, _ = base._grad_popargs(X, *args)
            for gg in atleast_tuple(g):
                if len(gg) == 0:
                    continue
                full = np.zeros((N, D) if gg.ndim < 3 else (N, D, gg.shape[2]))
                full[:, ends[i]:ends[i + 1]] = gg
                yield full

    def gram(self, X, y=None, *params, devices=None):
        """(Phi^T Phi, Phi^T y, y^T y) of the concatenation with Phi assembled ON the device: every
        child writes its column block of one feature matrix (random Fourier / FastFood / linear bases
        by kernels, anything else by one upload of its host block), one MFMA SYRK reduces it
        (slm.py:145-146,157 for a BasisCat).  f32 arithmetic -- so a concatenation with a ``dtype="f64"`` child
        (e.g. RandomLaplace, whose heavy-tailed W needs it) DECLINES (returns None) and the estimator takes the float64
        dense Gram of the transformed features instead."""
        if any(getattr(b, "dtype", "f32") != "f32" for b in self.bases):
            return None
        if devices is not None:  # row shards on several GPUs of this process (resident children), summed in HBM
            res = _sharded_gram(self, X, y, list(params), devices)
            if res is not None:
                return res
        N = X.shape[0]
        F = int(self.get_dim(X))
        ends = self.__base_locations(X)
        dev = _hip.get_device()
        chunk = int(max(32, min(N, (8 << 30) // (4 * ((F + 255) // 256 * 256)))))
        fm = _hip.FeatureMatrix(chunk, F)
        acc = dev.zeros((F * F + F + 1) * 8)
        base = acc.ptr.value
        pG, pb, pt = (_hip.ctypes.c_void_p(base), _hip.ctypes.c_void_p(base + F * F * 8),
                      _hip.ctypes.c_void_p(base + (F * F + F) * 8))
        for r0 in range(0, N, chunk):
            Xc = X[r0:r0 + chunk]
            fm.begin(Xc.shape[0])
            args = list(params)
            for i, b in enumerate(self.bases):
                args = b._put_features_popargs(Xc, fm, int(ends[i]), *args)
            dy = None if y is None else dev.upload_vector(np.ascontiguousarray(y[r0:r0 + chunk], dtype=np.float32))
            fm.gram_into(dy, pG, None if y is None else pb, None if y is None else pt)
            dev.sync()
        _hip._check(dev.lib, dev.lib.rr_symmetrize_dev(dev.ctx, pG, F))
        out = dev.download(acc, (F * F + F + 1,), np.float64)
        acc.free()
        G = out[:F * F].reshape(F, F)
        if y is None:
            return G, None, None
        return G, out[F * F:F * F + F].copy(), float(out[-1])

    def device_fit_state(self, X, y):
        """(X, y) resident for a whole fit when every child can take part (random Fourier / FastFood, Linear, Bias);
        None otherwise (the estimator then uses transform / grad)."""
        # a child that asks for float64 arithmetic end to end makes the whole state float64 (rr_featmat64: float64 feature
        # matrix, f64 MFMA Gram and second pass -- the reference's arithmetic, north star's 1e-5)
        dtype = "f64" if any(getattr(b, "dtype", "f32") == "f64" for b in self.bases) else "f32"
        children = []
        for b in self.bases:
            c = b._resident_child(X, dtype=dtype)
            if c is None:
                for done in children:
                    done.release()
                return None
            children.append(c)
        if not any(c.nparams for c in children):  # nothing to learn on the device
            for done in children:
                done.release()
            return None
        return CatFitState(self, children, X, y, dtype=dtype)

    def predict_moments(self, X, hypers, m, C):
        """(Phi m, rowsum((Phi C) o Phi)) with Phi assembled on the device (slm.py:240-243); None when a child
        asks for f64 arithmetic."""
        if any(getattr(b, "dtype", "f32") != "f32" for b in self.bases):
            return None
        N, F = X.shape[0], int(self.get_dim(X))
        ends = self.__base_locations(X)
        Fp = (F + 255) // 256 * 256
        # row chunks of at most 65 536: the feature matrix and its scratch (3 x chunk x Fp floats) are kept between calls
        # -- allocating 16 GB of them per call cost 0.6 s at N = 300 k, F = 4129, ten times the arithmetic
        chunk = int(max(256, min(N, 65536, (24 << 30) // (12 * Fp))))
        chunk = (chunk + 255) // 256 * 256
        cache, key = _handle_cache(self, "_pm_fm")
        fm = cache.get(key)
        if fm is None or fm.F != F or fm.max_rows < chunk:
            cache[key] = fm = None  # free the old one first
            fm = cache[key] = _hip.FeatureMatrix(chunk, F)
        fm.pass2_begin(m, C, predict=True)
        Ey, Vf = np.empty(N), np.empty(N)
        for r0 in range(0, N, chunk):
            Xc = X[r0:r0 + chunk]
            fm.begin(Xc.shape[0])
            args = list(atleast_list(hypers))
            for i, b in enumerate(self.bases):
                args = b._put_features_popargs(Xc, fm, int(ends[i]), *args)
            Ey[r0:r0 + chunk], Vf[r0:r0 + chunk] = fm.predict_rows(Xc.shape[0])
        return Ey, Vf

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_pm_fm", None)  # device feature matrix of predict_moments: per process, never pickled
        return state

    def get_dim(self, X):
        return np.sum(self.__all_dims(X))

    def params_values(self):
        return [v for b in self.bases for v in b.params_values()]

    @property
    def regularizer(self):
        return [b.regularizer for b in self.bases]

    def regularizer_diagonal(self, X, *regularizer):
        regularizer = repeat(None) if regularizer == () else regularizer
        regs, _ = zip(*(b.regularizer_diagonal(X, r) for b, r in zip(self.bases, regularizer)))
        if self.__slices is None:
            ends = self.__base_locations(X)
            self.__slices = [slice(b, e) for b, e in zip(ends[:-1], ends[1:])]
        return np.concatenate(regs), self.__slices

    @property
    def params(self):
        """All Parameter objects in concatenation order (basis_functions.py:1750-1763).  A child with SEVERAL parameters
        (FastFoodGM: mean and lenscale) contributes them one after the other, matching how `transform(X, *params)` routes
        them; the reference's version asks the child's LIST for `.has_value` and raises AttributeError, so a spectral
        mixture (several FastFoodGM components concatenated, as its docstring :1394-1396 recommends) cannot be fitted there."""
        plist = []
        for b in self.bases:
            for p in atleast_list(b.params):
                if p.has_value:
                    plist.append(p)
        if len(plist) == 0:
            return Parameter()
        return plist if len(plist) > 1 else plist[0]

    def __all_dims(self, X):
        if self.__dims is None:
            self.__dims = [b.get_dim(X) for b in self.bases]
        return self.__dims

    def __base_locations(self, X):
        if self.__baseinds is None:
            self.__baseinds = np.cumsum([0] + self.__all_dims(X))
        return self.__baseinds

    def __add__(self, other):
        if isinstance(other, BasisCat):
            return BasisCat(self.bases + other.bases)
        return BasisCat(self.bases + [other])

    def __radd__(self, other):
        return self if other == 0 else self.__add__(other)

    def __repr__(self):
        return "{}(basis_list={})".format(type(self).__name__, self.bases)
