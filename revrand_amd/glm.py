"""
The Bayesian generalised linear model with revrand's interface (reference: revrand/glm.py): a mixture of K
diagonal Gaussians over the weights, auto-encoding variational Bayes with stochastic gradients.

Same constructor, ``fit / predict / predict_moments / predict_logpdf / predict_cdf / predict_interval`` and
fitted attributes (``weights_, covariance_, regularizer_, like_hypers_, basis_hypers_``) as
``revrand.glm.GeneralizedLinearModel``.  Every O(M F L K) product of one SVI step (glm.py:205-322) runs on the
MI355X with Phi of the minibatch assembled in HBM and never returned to the host:

* ``Phi = basis.transform(X)``                           -> feature kernels into a device feature matrix  glm.py:215
* ``fs = ws.dot(Phi.T)`` for all K*L weight samples at once -> MFMA GEMM                                  glm.py:304
* ``likelihood.df / dp / loglike``                        -> one element-wise kernel with the reductions  glm.py:305,314,321
* ``Edws = dfs.dot(Phi)``, ``EdPhi = dfs.T.dot(ws)``      -> MFMA GEMMs                                   glm.py:308,311
* ``-(EdPhi * dPhi).sum()`` over ``basis.grad``           -> contraction kernel, no (M, F, d) tensor      glm.py:274-275

When the model is one random Fourier basis over minibatches gathered on the device, the whole SGD loop around these
products -- from_log, the O(F K^2) mixture-entropy terms, the gradient, the log trick's chain rule, the bounds and the
updater (optimize/sgd.py:337-425, decorators.py:329-408) -- runs with its parameters resident in HBM as well, one library
call per step and nothing read back until the fit ends (``_ResidentLoop``, rr_glm_sgd_*); otherwise those parts stay on
the host, as in the reference.  The standard-normal draws come from the host ``random_`` in the reference's order
(``randn(nsamples, D)`` per component), so a seeded run consumes the same random stream as the reference.
"""
import logging
import os
import time
from itertools import chain

import numpy as np
from scipy.stats.distributions import gamma, norm
from sklearn.base import BaseEstimator, RegressorMixin
from sklearn.utils import check_random_state
from sklearn.utils.validation import check_array, check_is_fitted, check_X_y

from . import _hip
from .basis_functions import LinearBasis, MinibatchFeatures
from .btypes import Bound, Parameter, Positive
from .likelihoods import Bernoulli, Binomial, Gaussian, Poisson
from .optimize import logtrick_sgd, sgd, structured_sgd
from .utils import atleast_list, issequence

log = logging.getLogger(__name__)


class _RowIndex(object):
    """`np.arange(N)` as far as the batch generator is concerned (`shape[0]`, fancy indexing) without the 65 536-row gather per
    step that `np.arange(N)[ind]` costs on the minibatch worker: `self[ind]` is `ind`."""

    def __init__(self, N):
        self.shape = (N,)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, ind):
        return ind


class _LazyRows(object):
    """Targets / per-row likelihood arguments of a resident fit as the batch generator sees them: `self[ind]` only NOTES the
    rows (`_PendingRows`); whoever handles the batch next -- the pipeline's upload stage, else the step -- gathers them
    (`_gathered`).  The thread that cuts the batches also draws the step's 1 024 000 normals at config 5 and is the slowest
    stage of the pipeline: the 0.4 ms gather per step is better spent elsewhere."""

    def __init__(self, arr):
        self.arr, self.shape = arr, arr.shape

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, ind):
        return _PendingRows(self.arr, ind)


class _PendingRows(object):
    def __init__(self, arr, ind):
        self.arr, self.ind = arr, ind

    def __len__(self):
        return len(self.ind)

    def get(self):
        return self.arr[self.ind]


def _gathered(items):
    return [b.get() if isinstance(b, _PendingRows) else b for b in items]


class _RowStub(object):
    """The zero-width stand-in for X of a resident fit: `self[ind]` is an (len(ind), 0) array."""

    def __init__(self, N):
        self.shape = (N, 0)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, ind):
        return np.empty((len(ind), 0))


WGTRND = norm()                  # sampling distribution over mixture weights        glm.py:40
COVRND = gamma(a=2, scale=0.5)   # sampling distribution over mixture covariances    glm.py:41
LOGITER = 500                    # SGD iterations between ELBO log lines             glm.py:42


class GeneralizedLinearModel(BaseEstimator, RegressorMixin):
    """Bayesian GLM; parameters as the reference (glm.py:45-139): ``likelihood, basis, K, maxiter, batch_size,
    updater, nsamples, nstarts, random_state``, plus

    sampler : "host" (default) | "device"
        Where the standard-normal draws of the reparameterisation trick are made.  "host": from ``random_`` in the
        reference's order, so a seeded run consumes the reference's random stream (1 M draws per step at config 5's
        shape: 18 ms, more than everything else together).  "device": a counter-based generator on the GPU keyed by
        one integer drawn from ``random_`` at the start of ``fit`` and the step number -- reproducible for a given
        seed, statistically equivalent, not the reference's stream; only the (D, K) expectations come back.
    distributed : bool
        Row-sharded SVI, one process per GPU under ``torch.distributed``: every rank calls ``fit`` with ITS rows and
        the same integer ``random_state``; per step each rank takes a minibatch of its shard, the Monte-Carlo sums
        ``[Edm | EdC | sum loglike | likelihood sums | basis gradient | batch rows]`` are all-reduced (one message,
        2 D K + O(d) numbers) and every rank applies the same update.  Over an RCCL communicator the optimiser's loop is
        resident on every rank and the sums are all-reduced in HBM (rr_glm_sgd_dist_step).
    devices : None | sequence of GPU indices | int | "all"
        Several GPUs behind THIS call, in this process (``revrand_amd.multigpu``): the rows of X stay resident sharded over
        the listed GPUs, every member serves the rows of a minibatch that fall into its shard, and the step's Monte-Carlo
        sums are added over the members -- in HBM, with the optimiser's loop resident on every member (rr_glm_sgd_group_step),
        or on the host when the fit is one the device loops do not cover.  The minibatch stream, the draws and the optimiser
        are the single-GPU run's; minibatches of fewer than 2048 rows per member are not split."""

    def __init__(self, likelihood=Gaussian(), basis=LinearBasis(), K=10, maxiter=3000, batch_size=10, updater=None,
                 nsamples=50, nstarts=500, random_state=None, sampler="host", distributed=False, gram_engine=None,
                 devices=None):
        self.likelihood = likelihood
        self.basis = basis
        self.K = K
        self.maxiter = maxiter
        self.batch_size = batch_size
        self.updater = updater
        self.nsamples = nsamples
        self.nstarts = nstarts
        self.random_state = random_state
        self.sampler = sampler
        self.distributed = distributed
        self.gram_engine = gram_engine  # arithmetic of the step's GEMMs (see StandardLinearModel); None = context setting
        self.devices = devices          # several GPUs behind this call, in this process (see StandardLinearModel; multigpu.py)
        self.random_ = check_random_state(self.random_state)

    def fit(self, X, y, likelihood_args=()):
        """Learn the posterior mixture and the hyper-parameters (glm.py:141-203)."""
        with self._engine_scope():
            return self._fit(X, y, likelihood_args)

    def _group(self):
        if getattr(self, "devices", None) is None:
            return None
        if self.distributed:
            raise ValueError("devices= (several GPUs in this process) and distributed=True (one process per GPU) cannot be combined")
        from . import multigpu
        return multigpu.get_group(self.devices)

    def _engine_scope(self):
        g = self._group()
        return _hip.gram_engine_scope(getattr(self, "gram_engine", None), None if g is None else g.members)

    def _fit(self, X, y, likelihood_args=()):
        X, y = check_X_y(X, y)
        self._drop_serving()
        self._dev_seed = None  # the device sampler is re-keyed from random_ per fit
        N, _ = X.shape
        self.B_ = X.shape[0] / self.batch_size
        if self.distributed:  # batch magnification of the whole job: N_total / (sum of the ranks' minibatch sizes)
            from . import parallel
            tot = parallel.allreduce_host(np.array([float(N), float(min(self.batch_size, N))]))
            self.B_ = tot[0] / tot[1]
        self.D_ = self.basis.get_dim(X)
        likelihood_args = _reshape_likelihood_args(likelihood_args, N)
        data = (X, y) + likelihood_args
        # when every basis can keep its columns of X in HBM, minibatches are gathered there: the optimiser then
        # shuffles row INDICES (the same permutation stream) and a zero-width stand-in for X
        self._resident_fit = self._features().make_resident(X)
        if self._resident_fit:
            lazy = tuple(_LazyRows(a) if (isinstance(a, np.ndarray) and len(a) == N) else a for a in (y,) + likelihood_args)
            data = (_RowStub(N),) + lazy + (_RowIndex(N),)
        params = [Parameter(WGTRND, Bound(), shape=(self.D_, self.K)),
                  Parameter(COVRND, Positive(), shape=(self.D_, self.K)),
                  self.basis.regularizer,
                  self.likelihood.params,
                  self.basis.params]
        loop = self._resident_loop(params, y, likelihood_args)
        fused = isinstance(getattr(loop, "_loop", loop), _FusedLoop)   # small minibatches: many steps per launch, nothing per batch on the device
        grouped = isinstance(loop, _GroupResidentLoop)   # the minibatch's rows spread over the members of a device group
        if loop is not None and not fused:  # (decided before the upload contexts' buffer rings are sized)
            self.__dict__["_draw_buffers"] = 6
            for f in (self._features().feats if grouped else [self._features()]):
                f.PREFETCH_SLOTS = 6
        log.info("Optimising parameters...")
        self.__it = -self.nstarts
        nsgd = structured_sgd(logtrick_sgd(sgd))
        if self.sampler == "device":
            # keyed here, so that `_elbo` never draws from random_ and the minibatches can be built one step ahead
            self._dev_seed, self._dev_step = int(self.random_.randint(0, 2 ** 31 - 1)), 0
        from functools import partial
        elbo = partial(GeneralizedLinearModel._elbo, self)
        elbo.objective_only = partial(GeneralizedLinearModel._elbo, self, objective_only=True)  # random starts
        sync = None
        if self.distributed:  # every rank starts from rank 0's candidates / start point
            from . import parallel
            comm = parallel.get_comm()
            sync = lambda v: comm.broadcast_host(v, root=0)  # noqa: E731
        # minibatch t+1 is built on a worker thread while step t runs on the device.  With the device sampler `_elbo` never
        # touches random_; with the reference's stream the worker also makes step t+1's draws right after cutting its
        # batch -- the order in which a sequential run consumes the stream -- and hands them over with the batch, so
        # the ~12 ms of randn per config-5 step overlap the kernels instead of preceding them.
        prefetch = self._ahead if (self.sampler == "device" or self._prefetch_draws) else False
        if fused:   # the worker thread cuts the batches and (reference's stream) makes the draws; the loop uploads them in blocks
            prefetch = [self._draw_ahead] if (self.sampler != "device" and self._prefetch_draws) else False
        # (RR_GLM_DRAW_UPLOAD=0: measurement switch, the step uploads its draws itself)
        if callable(prefetch) and not fused and self.sampler != "device" and self._prefetch_draws and self._native_draws \
                and getattr(self._features(), "accepts_device_draws", False) \
                and os.environ.get("RR_GLM_DRAW_UPLOAD", "1") != "0":  # the worker uploads its draws too: own context, three buffers in turn
            # (one upload context per process and device, shared by every fit: a context is a stream and two events, and a
            # cross-validation loop must not accumulate them)
            self.__dict__["_draw_upload"] = (_hip.get_upload_device(_hip.get_device().index), [None] * self._draw_buffers, [0])
        # (RR_GLM_BATCH_PREFETCH=0: measurement switch, the step uploads its indices / targets and gathers its rows itself)
        if callable(prefetch) and not fused and self._resident_fit and os.environ.get("RR_GLM_BATCH_PREFETCH", "1") != "0" \
                and self._group() is None:  # (a device group's members gather for themselves, on their own threads)
            self.__dict__["_batch_upload"] = _hip.get_upload_device(_hip.get_device().index)
        if callable(prefetch) and grouped:   # the same two hand-overs per MEMBER: an upload context on each member's GPU
            ups = [_hip.get_upload_device(d) for d in loop.group.devices]
            if os.environ.get("RR_GLM_BATCH_PREFETCH", "1") != "0":
                self.__dict__["_batch_upload"] = ups
            if self.sampler != "device" and self._prefetch_draws and self._native_draws and os.environ.get("RR_GLM_DRAW_UPLOAD", "1") != "0":
                self.__dict__["_draw_upload"] = [(u, [None] * self._draw_buffers, [0]) for u in ups]
        if callable(prefetch) and not fused and os.environ.get("RR_GLM_PREFETCH_STAGES", "2") != "1":
            # two workers in a row: the draws (2.5 ms of MT19937 + polar method per config-5 step, on the thread that cuts the
            # batches -- one RandomState, the reference's order -- next to its 0.4 ms of y[idx] and, once per epoch, 18 ms of
            # permutation(N)), then the uploads and gathers (1.1 ms): one worker doing both needed 4 ms per step, more than the
            # step's 3.4 ms of kernels under the resident loop.  With the device sampler the first stage only cuts the batches:
            # what it buys is the eight batches of slack behind it, which hide the permutation at an epoch boundary
            first = self._draw_ahead if self.sampler != "device" else (lambda batch: list(batch))
            prefetch = [first, lambda batch: self._ahead(batch, draws=False)]
        try:
            res = nsgd(elbo, params, data, eval_obj=True, maxiter=self.maxiter, updater=self.updater,
                       batch_size=self.batch_size, random_state=self.random_, nstarts=self.nstarts,
                       prefetch=prefetch, sync=sync, **({} if loop is None else {"device_loop": loop}))
        finally:
            # (sgd has stopped and joined its prefetch worker by now, also when `_elbo` raised; what the worker queued on the
            # upload context's stream must have landed before the buffers it wrote to are freed)
            upb = self.__dict__.pop("_batch_upload", None)
            up = self.__dict__.pop("_draw_upload", None)
            ups = [] if up is None else (up if isinstance(up, list) else [up])   # (a device group: one per member)
            upbs = [] if upb is None else (upb if isinstance(upb, list) else [upb])
            for ctx in {id(c): c for c in upbs + [u[0] for u in ups]}.values():
                try:
                    ctx.sync()
                except Exception:  # the original exception, if any, is the one to report
                    log.exception("synchronising the upload context failed")
            if loop is not None:   # (idempotent: whatever the loop still holds on the device -- also when the random starts raised)
                try:
                    loop.abort()
                except Exception:
                    log.exception("releasing the device loop failed")
            self._resident_fit = False
            self.__dict__.pop("_draw_buffers", None)
            self.__dict__.pop("_draw_ring", None)
            self._release_features()
            for u in ups:
                for buf in u[1]:
                    if buf is not None:
                        buf.free()
        (self.weights_, self.covariance_, self.regularizer_, self.like_hypers_, self.basis_hypers_) = res.x
        log.info("Finished! reg = {}, likelihood_hypers = {}, basis_hypers = {}, message: {}."
                 .format(self.regularizer_, self.like_hypers_, self.basis_hypers_, res.message))
        return self

    def _iteration(self, advance=0):
        """The SGD iteration counter `_elbo` logs by (glm.py:233-236), for the resident loop that runs instead of `_elbo`."""
        it = self.__it
        self.__it = it + advance
        return it

    _prefetch_draws = True  # False: `_elbo` draws for itself, strictly sequentially (what the tests compare against)
    _draw_buffers = 3
    _resident_sgd = True    # False (or RR_GLM_RESIDENT_SGD=0): always the host loop around `_elbo` (tests, A/B runs)

    _fused_sgd = True       # False (or RR_GLM_FUSED=0): never the many-steps-per-launch loop of small minibatches

    def _resident_loop(self, params, y=None, likelihood_args=()):
        """The SGD loop with parameters, updater state and gradient in device memory (`_ResidentLoop`, rr_glm_sgd) when this
        fit is one it covers: minibatches gathered on the device; the basis a random Fourier basis, a FastFoodRBF or FastFoodGM
        (through the dense equivalent of its chain), a LinearBasis or a concatenation of such children (Xdim <= 128, a scalar
        regulariser each); one of
        the reference's likelihoods and updaters; K <= 64 (the fused small-batch loop: K <= 32); one process; one GPU, or --
        `devices=` -- every member of the device group (`_GroupResidentLoop`), or member 0 alone when the minibatches are too small
        to split.  None otherwise -- the host loop around `_elbo` then runs, with the same results."""
        from . import optimize as opt
        if not self._resident_sgd or os.environ.get("RR_GLM_RESIDENT_SGD", "1") == "0":
            return None
        feats = self._features()
        if not getattr(self, "_resident_fit", False) or self.sampler not in ("host", "device"):
            return None
        if self.distributed:
            # one process per GPU: the step-per-call loop with its two all-reduces over the ranks in HBM (rr_glm_sgd_dist_step)
            # when the job's communicator is RCCL on this rank's context; a gloo group (the CPU test transport) keeps the
            # host loop, whose one all-reduce per step goes through the host
            from . import parallel
            comm = parallel.get_comm()
            if type(feats) is not MinibatchFeatures:
                return None
            if getattr(comm, "world", 1) == 1:
                return self._one_device_loop(feats, params, y, likelihood_args)
            if getattr(comm, "kind", None) != "rccl" or comm.dev.ctx.value != _hip.get_device().ctx.value:
                return None
            return self._one_device_loop(feats, params, y, likelihood_args, comm=comm.h)
        group = self._group()
        if group is not None:
            # a device group: the minibatch's rows spread over ALL members (`_GroupResidentLoop`: per-member products, the row
            # sums all-reduced in HBM, the update replicated), or -- minibatches too small to split (multigpu.
            # ShardedMinibatchFeatures.n_use == 1: the reference's default of 10 rows) -- the one-device loops on member 0
            from .multigpu import ShardedMinibatchFeatures
            if type(feats) is not ShardedMinibatchFeatures or not feats.resident or feats.n_use not in (1, group.n):
                return None
            if feats.n_use == 1:
                with _hip.device_scope(group.members[0]):
                    inner = self._one_device_loop(feats.feats[0], params, y, likelihood_args)
                return None if inner is None else _ScopedLoop(inner, group.members[0])
            kids = [self._loop_children(f, params) for f in feats.feats]
            if any(k is None for k in kids) or not self._loop_covers(params):
                return None
            return _GroupResidentLoop(self, feats, self._n_lik(params), kids)
        if type(feats) is not MinibatchFeatures:
            return None
        return self._one_device_loop(feats, params, y, likelihood_args)

    def _n_lik(self, params):
        return sum(int(np.prod(p.shape, dtype=int)) for p in atleast_list(params[3]))

    def _loop_covers(self, params):
        """updater, likelihood, mixture size and likelihood parameters are ones the device loops implement"""
        from . import optimize as opt
        if not 1 <= self.K <= 64:
            return False
        if self.updater is not None and type(self.updater) not in (opt.SGDUpdater, opt.AdaDelta, opt.AdaGrad, opt.Momentum, opt.Adam):
            return False
        if type(self.likelihood) not in (Bernoulli, Binomial, Gaussian, Poisson):
            return False
        return self._n_lik(params) == (1 if type(self.likelihood) is Gaussian else 0)

    def _loop_children(self, feats, params):
        """The children of one device's MinibatchFeatures as rr_glm_sgd takes them, or None when one is not covered."""
        from .basis_functions import _ResidentFastFood, _ResidentFastFoodGM, _ResidentLinear, _ResidentRFF
        kids = getattr(feats, "_kids", [])
        if not 1 <= len(kids) <= 16:
            return None
        regs = atleast_list(params[2])
        if len(regs) != len(kids) or any(getattr(p, "shape", None) != () for p in regs):  # one scalar regulariser per child
            return None
        children = []
        for kid, b in zip(kids, feats.bases):
            # (a FastFoodRBF child: through its dense equivalent W = _makeVX(I_d) -- the chain is linear in x, so
            # [cos | sin](x (W / l)) / sqrt(n) IS basis_functions.py:1263-1289 -- on the random Fourier kernels: its handle is
            # that basis; the chain kernel stays the host loop's and the SLM's route)
            if type(kid) in (_ResidentRFF, _ResidentFastFood) and b.d <= 128 and kid.W.shape[0] == b.d:
                n_ls = int(np.prod(b.params.shape, dtype=int))
                if n_ls not in (1, b.d):
                    return None
                children.append(("rff", kid.h, n_ls))
            elif type(kid) is _ResidentFastFoodGM and b.d <= 128 and kid.W.shape[0] == b.d \
                    and [getattr(p, "shape", None) for p in atleast_list(b.params)] == [(b.d,), (b.d,)]:
                # (a spectral-mixture component: [cos | sin](VX + mX) | [cos | sin](VX - mX) are two random Fourier blocks of
                # the chain's dense equivalent with every frequency moved by +- mean -- basis_functions.py:1443-1475 -- and its
                # two gradients two sums over their contractions: _ResidentFastFoodGM.dhyp)
                children.append(("gm", kid.h, 2 * b.d))
            elif type(kid) is _ResidentLinear and int(np.prod(b.params.shape, dtype=int)) == 0:
                children.append(("linear", int(kid.dX.shape[1]), bool(kid.onescol)))
            else:
                return None
        n_par = sum(int(np.prod(getattr(p, "shape", ()), dtype=int)) for p in _flat_params(params[4]))
        if sum(c[2] for c in children if c[0] in ("rff", "gm")) != n_par:
            return None
        return children

    def _one_device_loop(self, feats, params, y, likelihood_args, comm=None):
        if not self._loop_covers(params):
            return None
        children = self._loop_children(feats, params)
        if children is None:
            return None
        kids, n_lik = feats._kids, self._n_lik(params)
        # small minibatches (the reference's default is 10 rows): the whole loop inside one kernel, many steps per launch
        # (not between ranks: that kernel has no exchange step)
        if self._fused_sgd and os.environ.get("RR_GLM_FUSED", "1") != "0" and y is not None and len(likelihood_args) <= 1 \
                and np.isfinite(self.maxiter) and comm is None and all(c[0] != "gm" for c in children):
            N = len(y)
            M = int(min(self.batch_size, N))
            F = int(self.D_)
            dsum = sum((kid.W.shape[0] if c[0] == "rff" else c[1]) for kid, c in zip(kids, children))
            n_ls = sum(c[2] for c in children if c[0] == "rff")
            if _hip.svi_supported(F, self.K, self.nsamples, M, len(children), dsum, n_ls):
                return _FusedLoop(self, feats, n_lik, children, y, likelihood_args)
        return _ResidentLoop(self, feats, n_lik, children, comm=comm)

    def _reference_draws(self, out=None):
        """The step's standard normals from `random_` in the reference's order (glm.py:300): randn(L, D) per component.
        out: a (K L, D) float32 array to fill (the minibatch pipeline's ring, `_draw_ahead`)."""
        K, L_, D = self.K, self.nsamples, self.D_
        if self._native_draws:  # the same K consecutive randn(L, D) as one call of the library's generator (bit-identical)
            return _hip.legacy_randn(self.random_, K * L_ * D, np.float32, out=out).reshape(K * L_, D)
        e = np.empty((K * L_, D), dtype=np.float32) if out is None else out
        for k in range(K):
            e[k * L_:(k + 1) * L_] = self.random_.randn(L_, D)
        return e

    _native_draws = True  # False: NumPy generates the draws (what tests compare the library's generator against)

    def _ahead(self, batch, draws=True):
        """On the minibatch worker thread, for the batch it just cut: the likelihood's constants of that batch when they do
        not depend on parameters (Poisson / binomial log-factorial sums: 0.15 ms of a 5 ms config-5 step), and -- with the
        reference's random stream -- the step's draws (`_draw_ahead`)."""
        batch = _gathered(batch)
        made = batch.pop() if (batch and isinstance(batch[-1], _Draws)) else None   # (by an earlier pipeline stage)
        resident = getattr(self, "_resident_fit", False)
        spec = None
        if getattr(self.likelihood, "spec_is_parameter_free", False):
            extra = batch[2:-1] if resident else batch[2:]  # likelihood arguments of the batch
            spec = self.likelihood.device_spec(batch[1], [], extra)
        # resident data: the batch's index upload, row gathers and target upload happen here too, on the upload context's
        # stream, while an earlier step runs (the per-row likelihood argument is known ahead unless it depends on parameters)
        gathered = None
        up = self.__dict__.get("_batch_upload")
        if resident and up is not None and (spec is not None or isinstance(self.likelihood, Gaussian)):
            rowarg = spec[2] if spec is not None else None
            gathered = _Batch(self._features().prefetch_batch(up, batch[-1], batch[1], rowarg))
        if spec is not None:
            batch.append(_Spec(spec))
        if gathered is not None:
            batch.append(gathered)
        if made is not None:
            batch.append(self._upload_draws(made))
        elif draws and self.sampler != "device" and self._prefetch_draws:
            batch = self._draw_ahead(batch, upload=True)
        return batch

    def _draw_ahead(self, batch, upload=False):
        """On a minibatch worker thread, right after the batch is cut: the step's draws from `random_` in the reference's
        order (batch_t, e_t, batch_t+1, ...); `upload`: also `_upload_draws` (a single-stage pipeline)."""
        # (into one of 16 host arrays in turn: fewer than that many batches are alive between this stage and the step that
        # consumes them -- eight in the queue behind it, one in each later stage and queue -- and 4 MB arrays allocated and
        # freed per step are mapped and unmapped by the allocator: `_hip.legacy_randn`)
        ring = self.__dict__.setdefault("_draw_ring", [[], 0])
        shape = (self.K * self.nsamples, self.D_)
        if len(ring[0]) < 16 or ring[0][0].shape != shape:
            if ring[0] and ring[0][0].shape != shape:
                ring[0].clear()
            ring[0].append(np.empty(shape, dtype=np.float32))
            out = ring[0][-1]   # (every entry is written in full by the draw that follows: touched at once)
        else:
            out = ring[0][ring[1] % 16]
        ring[1] += 1
        d = _Draws(self._reference_draws(out))
        return list(batch) + [self._upload_draws(d) if upload else d]

    def _upload_draws(self, d):
        """On a minibatch worker thread: the draws' upload -- through a device context (stream) of this
        thread's own, into one of THREE buffers in turn: the worker runs up to two steps ahead (one finished batch waits in
        the queue while the next is being made) -- while earlier steps' kernels run; the step then starts from
        device-resident draws (rr_featmat_glm_step_draws_dev).  (SIX buffers under the resident loop, whose host queues
        steps up to two ahead of the device on top of that: `_draw_buffers`.)"""
        up = self.__dict__.get("_draw_upload")
        if up is None:
            return d
        e = d.e

        def upload(dev, bufs, turn):
            buf = bufs[turn[0] % len(bufs)]
            if buf is None or buf.nbytes < e.nbytes:
                if buf is not None:
                    buf.free()
                buf = bufs[turn[0] % len(bufs)] = dev.malloc(e.nbytes)
            _hip._check(dev.lib, dev.lib.rr_memcpy_h2d(dev.ctx, buf.ptr, e.ctypes.data_as(_hip.ctypes.c_void_p), e.nbytes))
            buf.shape, buf.dtype = e.shape, e.dtype
            turn[0] += 1
            return buf
        if isinstance(up, list):   # a device group: the same draws into every member's HBM
            return _Draws(tuple(upload(*u) for u in up))
        return _Draws(upload(*up))

    # -- device features of one minibatch ------------------------------------------------------
    def _features(self):
        f = self.__dict__.get("_mbf")
        if f is None:
            g = self._group()
            if g is not None:
                from . import multigpu
                f = multigpu.ShardedMinibatchFeatures(self.basis, g, batch_size=self.batch_size)
            else:
                f = MinibatchFeatures(self.basis)
            self.__dict__["_mbf"] = f
        return f

    def _release_features(self):
        f = self.__dict__.pop("_mbf", None)
        if f is not None:
            f.release()

    def __getstate__(self):
        state = dict(super().__getstate__())  # sklearn's (adds its version tag)
        state.pop("_mbf", None)
        state.pop("_draw_buffers", None)
        state.pop("_draw_ring", None)
        state.pop("_resident_clock", None)
        state.pop("_serve_feats", None)
        state.pop("_draw_upload", None)
        state.pop("_batch_upload", None)
        return state

    def _drop_serving(self):
        srv = self.__dict__.pop("_serve_feats", None)
        if srv is not None and srv[0] == os.getpid():
            srv[1].release()

    def _elbo(self, m, C, reg, lpars, bpars, X, y, *largs, objective_only=False):
        """-ELBO and its gradients on one minibatch (glm.py:205-294).  objective_only (the random starts of `fit`):
        -ELBO alone, from fs and the likelihood sums -- none of the gradient GEMMs, no basis gradients."""
        D, K = m.shape
        L_ = self.nsamples
        lpars_l = atleast_list(lpars)
        if isinstance(y, _PendingRows) or any(isinstance(a, _PendingRows) for a in largs):  # (no pipeline stage gathered them)
            y, largs = y.get() if isinstance(y, _PendingRows) else y, tuple(_gathered(largs))

        draws, spec = None, None
        gathered = None
        while largs and isinstance(largs[-1], (_Draws, _Spec, _Batch)):      # made ahead on the worker (`_ahead`)
            if isinstance(largs[-1], _Draws):
                draws = largs[-1].e
            elif isinstance(largs[-1], _Batch):
                gathered = largs[-1].token
            else:
                spec = largs[-1].spec
            largs = largs[:-1]
        feats = self._features()
        resident = getattr(self, "_resident_fit", False)
        if resident:                                                          # rows by index from the resident data
            idx, largs = largs[-1], largs[:-1]
        # everything the step needs from the host goes first (likelihood constants, the targets' upload): the feature
        # kernels launched next then run while the host gets to the step's own call, instead of being waited for
        lid, lpar, rowarg, llconst = spec if spec is not None else self.likelihood.device_spec(y, lpars_l, largs)
        if resident and gathered is not None:
            if not feats.take_prefetched_targets(gathered, y, rowarg):
                feats.stage_targets(y, rowarg)
            feats.assemble_idx(idx, atleast_list(bpars), gathered)
        elif resident:
            feats.stage_targets(y, rowarg)
            feats.assemble_idx(idx, atleast_list(bpars))
        else:
            feats.assemble(X, atleast_list(bpars))                            # Phi (M x D) in HBM
        objective_only = objective_only and getattr(feats, "supports_objective_only", False)
        # the mixture-entropy terms depend on (m, C) only: a worker thread forms them while the device call below (which
        # releases the GIL) runs the step's kernels -- ~0.7 ms of a 6 ms config-5 step
        # (with NumPy generating the reference's stream the step waits for the randn worker, and a second helper thread
        # only takes the GIL away from it; the library's generator holds no GIL)
        helper = self.sampler == "device" or self._native_draws
        mixture = _submit(_mixture_terms, m, C) if (helper and not objective_only) else None
        okw = {"objective_only": True} if objective_only else {}
        if self.sampler == "device":
            if self.__dict__.get("_dev_seed") is None:
                self._dev_seed, self._dev_step = int(self.random_.randint(0, 2 ** 31 - 1)), 0
            Edm, EdC, llsum, aux = feats.glm_step_sampled(y, rowarg, lid, lpar, m, C, K, L_, self._dev_seed,
                                                          self._dev_step, **okw)
            self._dev_step += 1
        else:
            if self.sampler != "host":
                raise ValueError("sampler must be 'host' or 'device'")
            # the reference's draws, in its order (glm.py:300): randn(L, D) per component; ws = m_k + sqrt(C_k) e and
            # the reductions over the samples (glm.py:309-310) happen on the device
            if draws is None:
                self.D_ = D
                draws = self._reference_draws()
            Edm, EdC, llsum, aux = feats.glm_step_draws(y, rowarg, lid, lpar, m, C, K, L_, draws, **okw)

        nrows = float(len(y))
        if objective_only:
            if self.distributed:
                from . import parallel
                buf = parallel.allreduce_host(np.concatenate((llsum, [llconst])))
                llsum, llconst = buf[:K], float(buf[K])
            L, slices = self.basis.regularizer_diagonal(X, *atleast_list(reg))
            iL = 1. / L[:, np.newaxis]
            logNkl = _qmatrix(m, C)
            mx = logNkl.max(axis=0)
            logzk = np.log(np.exp(logNkl - mx).sum(axis=0)) + mx
            Ell = llsum / L_ + llconst
            ELBO = (Ell.sum() * self.B_ - 0.5 * D * K * np.log(2 * np.pi) - 0.5 * K * np.log(L).sum()
                    - 0.5 * ((m ** 2 + C) * iL).sum() - logzk.sum() + np.log(K)) / K
            self.__it += 1
            return -ELBO
        # -(EdPhi o dPhi).sum() per parameter: a contraction kernel + a small download.  On the helper thread (the call
        # holds no GIL) while this thread does the step's NumPy arithmetic below; joined where the value is needed
        dbp = _submit(feats.glm_basis_grads, X) if (helper and not self.distributed) else None
        dbpars = feats.glm_basis_grads(X) if dbp is None else None
        if self.distributed:  # one exchange per step: the per-rank Monte-Carlo sums
            from . import parallel
            from .utils import flatten_values
            flat_db = flatten_values(dbpars)
            buf = np.concatenate((Edm.ravel(), EdC.ravel(), llsum, aux, flat_db, [llconst, nrows]))
            buf = parallel.allreduce_host(buf)
            o = 0
            Edm = buf[o:o + D * K].reshape(D, K); o += D * K
            EdC = buf[o:o + D * K].reshape(D, K); o += D * K
            llsum = buf[o:o + K]; o += K
            aux = buf[o:o + K]; o += K
            red_db = buf[o:o + flat_db.size]; o += flat_db.size
            llconst, nrows = float(buf[o]), float(buf[o + 1])
            dbpars = _like_structure(dbpars, red_db)

        L, slices = self.basis.regularizer_diagonal(X, *atleast_list(reg))
        iL = 1. / L[:, np.newaxis]
        logzk, mix_m, mix_C = mixture.result() if mixture is not None else _mixture_terms(m, C)

        dlpars = [np.zeros_like(p) for p in lpars_l]
        dolog = (self.__it % LOGITER == 0) or (self.__it == self.maxiter - 1)
        calc_ll = dolog or (self.__it < 0)
        Ell = llsum / L_ + llconst

        dm = (self.B_ * Edm - m / L[:, np.newaxis] + mix_m) / K
        dC = (self.B_ * EdC - 1. / L[:, np.newaxis] + mix_C) / (2 * K)
        if len(dlpars) > 0:  # only the Gaussian has a likelihood parameter: dp = ((y-f)^2/var^2 - 1/var)/2
            ivar = 1. / lpar
            Edlp = 0.5 * (aux * ivar ** 2 - ivar * nrows * L_) / L_
            dlpars[0] = dlpars[0] - Edlp.sum() / K

        def dreg(s):
            return -0.5 * (((m[s] ** 2 + C[s]) * iL[s] ** 2).sum() / K - iL[s].sum())

        dL = list(map(dreg, slices)) if issequence(slices) else dreg(slices)

        ELBO = -np.inf
        if calc_ll:
            ELBO = (Ell.sum() * self.B_ - 0.5 * D * K * np.log(2 * np.pi) - 0.5 * K * np.log(L).sum()
                    - 0.5 * ((m ** 2 + C) * iL).sum() - logzk.sum() + np.log(K)) / K
        if dolog:
            log.info("{}Iter {}: ELBO = {}, reg = {}, like_hypers = {}, basis_hypers = {}"
                     .format("Random starts: " if self.__it < 0 else "", self.__it, ELBO, reg, lpars, bpars))
        self.__it += 1
        if dbp is not None:
            dbpars = dbp.result()
        return -ELBO, [-dm, -dC, dL, dlpars, dbpars]

    # -- prediction -----------------------------------------------------------------------------
    def predict(self, X, nsamples=200, likelihood_args=()):
        """Expected target values (glm.py:324-349)."""
        Ey, _ = self.predict_moments(X, nsamples, likelihood_args)
        return Ey

    def predict_moments(self, X, nsamples=200, likelihood_args=()):
        """Monte-Carlo predictive mean and variance (glm.py:351-393)."""
        fs = self._sample_matrix(X, nsamples)                                 # N x nsamples
        Eyargs = tuple(chain(atleast_list(self.like_hypers_), likelihood_args))
        ys = np.empty(fs.shape)
        for i in range(nsamples):
            ys[:, i] = self.likelihood.Ey(fs[:, i], *Eyargs)
        Ey = ys.mean(axis=1)
        Vy = ((ys - Ey[:, np.newaxis]) ** 2).mean(axis=1)
        return Ey, Vy

    def predict_logpdf(self, X, y, nsamples=200, likelihood_args=()):
        """Mean / min / max log predictive density over the latent samples (glm.py:395-444)."""
        X, y = check_X_y(X, y)
        fs = self._sample_matrix(X, nsamples)
        llargs = tuple(chain(atleast_list(self.like_hypers_), likelihood_args))
        ps = np.empty(fs.shape)
        for i in range(nsamples):
            ps[:, i] = self.likelihood.loglike(y, fs[:, i], *llargs)
        return ps.mean(axis=1), ps.min(axis=1), ps.max(axis=1)

    def predict_cdf(self, X, quantile, nsamples=200, likelihood_args=()):
        """Predictive CDF at `quantile` (glm.py:446-495)."""
        fs = self._sample_matrix(X, nsamples)
        cdfarg = tuple(chain(atleast_list(self.like_hypers_), likelihood_args))
        ps = np.empty(fs.shape)
        for i in range(nsamples):
            ps[:, i] = self.likelihood.cdf(quantile, fs[:, i], *cdfarg)
        return ps.mean(axis=1), ps.min(axis=1), ps.max(axis=1)

    def predict_interval(self, X, percentile, nsamples=200, likelihood_args=(), multiproc=True):
        """Central `percentile` interval of the predictive distribution, (lower, upper) per query row -- the quantiles
        of the CDF averaged over the latent samples (glm.py:497-570).  The reference root-finds row by row in a process
        pool; here every row is bisected at once on the (N, nsamples) sample matrix (`multiproc` is accepted and
        ignored).  Same search bracket, and NaN where the bracket does not contain the quantile."""
        fs = self._sample_matrix(X, nsamples)                                 # N x nsamples
        N = fs.shape[0]
        rowargs = [np.asarray(a, dtype=float).reshape(N, 1) for a in _reshape_likelihood_args(likelihood_args, N)
                   if len(a)]
        largs = atleast_list(self.like_hypers_) + rowargs
        centre = self.likelihood.Ey(fs, *largs).mean(axis=1)
        reach = 1000. * np.maximum(centre, 1.)

        def sampled_cdf(q):
            return self.likelihood.cdf(q[:, np.newaxis], fs, *largs).mean(axis=1)

        tail = 0.5 * (1. - percentile)
        return _bisect_quantile(sampled_cdf, tail, reach), _bisect_quantile(sampled_cdf, 1. - tail, reach)

    def _sample_matrix(self, X, nsamples):
        """Latent function samples f = Phi w, (N, nsamples), the product on the device (glm.py:572-620)."""
        check_is_fitted(self, ['weights_', 'covariance_', 'basis_hypers_', 'like_hypers_', 'regularizer_'])
        X = check_array(X)
        D, K = self.weights_.shape
        k = self.random_.randint(0, K, size=(nsamples,))
        w = self.weights_[:, k] + self.random_.randn(D, nsamples) * np.sqrt(self.covariance_[:, k])
        # the feature matrix is kept between prediction calls (per process; dropped by fit and by pickling)
        srv = self.__dict__.get("_serve_feats")
        if srv is None or srv[0] != os.getpid():
            g = self._group()
            if g is not None:
                from . import multigpu
                served = multigpu.ShardedMinibatchFeatures(self.basis, g)
            else:
                served = MinibatchFeatures(self.basis)
            srv = self.__dict__["_serve_feats"] = (os.getpid(), served)
        return srv[1].project(X, atleast_list(self.basis_hypers_), w)

    def _sample_func(self, X, nsamples, genaxis=1):
        """Generator over latent function samples, column-wise (genaxis=1) or per observation (genaxis=0)."""
        if genaxis not in (0, 1):
            raise ValueError("Invalid axis to generate samples from")
        fs = self._sample_matrix(X, nsamples)
        return (f for f in (fs.T if genaxis == 1 else fs))

    def __repr__(self):
        return "{}(likelihood={}, basis={}, K={}, maxiter={}, batch_size={},updater={}, nsamples={}, nstarts={}, " \
            "random_state={})".format(type(self).__name__, self.likelihood, self.basis, self.K, self.maxiter,
                                      self.batch_size, self.updater, self.nsamples, self.nstarts, self.random_state)


class GeneralisedLinearModel(GeneralizedLinearModel):
    """GB/AU spelling (glm.py:640-642)."""


def _mixture_terms(m, C):
    """(log z_k, and the mixture parts of dm, dC) of the reference's loop over the K components (glm.py:238-262), all
    components at once: alpha[k, l] = N_kl / z_k + N_kl / z_l (glm.py:244-246)."""
    logNkl = _qmatrix(m, C)
    mx = logNkl.max(axis=0)
    logzk = np.log(np.exp(logNkl - mx).sum(axis=0)) + mx
    alpha = np.exp(logNkl.T - logzk[:, np.newaxis]) + np.exp(logNkl.T - logzk[np.newaxis, :])
    mkmj = m[:, :, np.newaxis] - m[:, np.newaxis, :]                         # D x k x l
    iCkCj = 1. / (C[:, :, np.newaxis] + C[:, np.newaxis, :])
    return (logzk, np.einsum("dkl,kl->dk", iCkCj * mkmj, alpha),
            np.einsum("dkl,kl->dk", iCkCj - (mkmj * iCkCj) ** 2, alpha))


_pool = None


def _submit(fn, *args):
    """Run fn(*args) on this process's helper thread (created on first use; never inherited across a fork)."""
    global _pool
    if _pool is None or _pool[0] != os.getpid():
        from concurrent.futures import ThreadPoolExecutor
        _pool = (os.getpid(), ThreadPoolExecutor(max_workers=1, thread_name_prefix="revrand-glm"))
    return _pool[1].submit(fn, *args)


class _ResidentLoop(object):
    """`optimize.sgd`'s loop (sgd.py:337-425) for GeneralizedLinearModel.fit with everything resident: the flat vector
    [m | C | reg | likelihood parameter | length scales] of structured_sgd (log space on the Positive coordinates, set by
    logtrick_sgd as `log_coordinates`), the updater's state and the gradient live in HBM; a step is ONE library call that
    queues from_log, the features, the SVI step, the mixture terms, the gradient, the chain rule, the bounds and the update
    (rr_glm_sgd_step) and returns without waiting -- the host's part of a step is taking the next minibatch (made on the
    worker thread: indices, gathered rows, targets and draws already on the device) off the queue.  `_elbo` is not called;
    what it would have logged every LOGITER iterations is logged from the device's objective."""

    log_coordinates = None
    comm = None   # distributed=True over RCCL: this rank's rr_comm -- the step's row sums are all-reduced over the ranks in HBM

    def __init__(self, glm, feats, n_lik, children, comm=None):
        self.glm, self.feats, self.n_lik, self.children = glm, feats, n_lik, children
        self.sgd = None
        self.comm = comm

    def begin(self, z0, lower, upper, updater, maxiter):
        from . import optimize as opt
        g, feats = self.glm, self.feats
        updater = opt.Adam() if updater is None else updater
        kind = type(updater).__name__
        par = {"SGDUpdater": lambda u: [u.eta], "AdaDelta": lambda u: [u.rho, u.epsilon], "AdaGrad": lambda u: [u.eta, u.epsilon],
               "Momentum": lambda u: [u.rho, u.eta], "Adam": lambda u: [u.alpha, u.beta1, u.beta2, u.epsilon]}[kind](updater)
        if not np.isfinite(maxiter):
            raise ValueError("the resident loop needs a finite maxiter")
        pos = self.log_coordinates if self.log_coordinates is not None else np.zeros(len(z0), dtype=bool)
        self.pos = np.asarray(pos, dtype=bool)
        self.t = 0
        self.clock = []   # host time at which each step was queued (the queue is two deep: it follows the device's pace)
        self._make = lambda M, fm=None, children=None: _hip.ResidentSgd(
            feats.fm if fm is None else fm, self.children if children is None else children, g.K, self.n_lik, z0, lower, upper,
            self.pos, _hip.UPDATER_IDS[kind], par, max(1, int(maxiter)))
        self._z0 = np.array(z0, dtype=float)

    def _start(self, M):
        """with the first minibatch: the feature matrix of M rows and the device loop on it"""
        self.feats._ensure(M, int(self.glm.D_))
        self.sgd = self._make(M)

    def _values(self):
        """(reg, likelihood parameters, basis parameters) as the host loop's log line prints them"""
        z = self.sgd.read()[0]
        x = np.where(self.pos, np.exp(np.where(self.pos, z, 0.0)), z)
        o, nk = 2 * self.glm.D_ * self.glm.K, len(self.children)
        regs = list(x[o:o + nk])
        ls, q = [], o + nk + self.n_lik
        for c in self.children:
            n = c[2] if c[0] in ("rff", "gm") else 0
            ls.append(x[q] if n == 1 else ([x[q:q + n // 2], x[q + n // 2:q + n]] if c[0] == "gm" else x[q:q + n]))
            q += n
        return (regs[0] if nk == 1 else regs), ([x[o + nk]] if self.n_lik else []), (ls[0] if nk == 1 else ls)

    def step(self, batch):
        g, feats = self.glm, self.feats
        batch = _gathered(batch)
        y, largs = batch[1], list(batch[2:])
        draws = spec = gathered = None
        while largs and isinstance(largs[-1], (_Draws, _Spec, _Batch)):      # made ahead on the worker (`_ahead`)
            last = largs.pop()
            if isinstance(last, _Draws):
                draws = last.e
            elif isinstance(last, _Batch):
                gathered = last.token
            else:
                spec = last.spec
        idx = largs.pop()
        if self.n_lik:   # Gaussian: the variance is a coordinate of z, its constant is formed on the device
            from .likelihoods import RR_LIK_GAUSSIAN
            lid, rowarg, llconst = RR_LIK_GAUSSIAN, None, 0.0
        else:
            lid, _, rowarg, llconst = spec if spec is not None else g.likelihood.device_spec(y, [], largs)
        it = g._iteration(advance=1)
        dolog = (it % LOGITER == 0) or (it == g.maxiter - 1)
        if self.sgd is None:
            self._start(len(idx))
        shown = self._values() if dolog else None   # (waits for the queue: twice per LOGITER steps)
        if gathered is not None and gathered.key == (id(y), len(y), rowarg is None):  # uploaded by the worker with the rows
            dy, dn = gathered.dy, gathered.dn
        else:
            dy, dn = feats._stage("y", y, np.float32), None if rowarg is None else feats._stage("rowarg", rowarg, np.float32)
        dX = feats.batch_rows(idx, gathered)
        dE, seed, key = None, 0, 0
        if g.sampler == "device":
            seed, key = g._dev_seed, g._dev_step
            g._dev_step += 1
        else:
            if draws is None:
                draws = g._reference_draws()
            dE = draws if isinstance(draws, _hip.DeviceBuffer) else feats._stage("E", draws, np.float32)
        self.sgd.step(dX, len(idx), dy, dn, lid, llconst, g.B_, g.nsamples, dE, seed, key, comm=self.comm)
        self.clock.append(time.perf_counter())
        if dolog:
            log.info("Iter {}: ELBO = {}, reg = {}, like_hypers = {}, basis_hypers = {}"
                     .format(it, -self.sgd.objective(self.t), shown[0], shown[1], shown[2]))
        self.t += 1

    def end(self):
        if self.sgd is None:   # no step was taken (maxiter = 0)
            return self._z0, np.empty(0), np.empty(0)
        z, objs, norms = self.sgd.read()
        self.clock.append(time.perf_counter())
        self.glm.__dict__["_resident_clock"] = np.array(self.clock)   # (measurement: bench.py, tools/c5_resident.py)
        self.abort()
        return z, objs, norms

    def abort(self):
        sgd, self.sgd = self.sgd, None
        if sgd is not None:
            sgd.close()
        self.feats.__dict__.pop("PREFETCH_SLOTS", None)


class _ScopedLoop(object):
    """A one-device loop (`_ResidentLoop`, `_FusedLoop`) on ONE member of a device group -- `GeneralizedLinearModel(devices=...)`
    with minibatches too small to split: every call runs with the member's context as the calling thread's default device
    (`_hip.device_scope`), everything else is the inner loop's."""

    def __init__(self, loop, dev):
        object.__setattr__(self, "_loop", loop)
        object.__setattr__(self, "_dev", dev)

    def __getattr__(self, name):
        v = getattr(self._loop, name)
        if not callable(v):
            return v

        def scoped(*args, **kwargs):
            with _hip.device_scope(self._dev):
                return v(*args, **kwargs)
        return scoped

    def __setattr__(self, name, value):
        setattr(self._loop, name, value)


class _GroupResidentLoop(_ResidentLoop):
    """`_ResidentLoop` over the members of a device group (`GeneralizedLinearModel(devices=[...])`, multigpu.DeviceGroup): X's
    rows are sharded over the members (multigpu.ShardedMinibatchFeatures), a minibatch -- the optimiser's own index stream --
    is split by owner, every member gathers ITS rows and runs the step's products on them, the three row sums (the
    length-scale contractions, [Edm | EdC], the likelihood sums) are all-reduced in HBM and every member makes the same update
    of its own copy of the flat vector (rr_glm_sgd_group_step: the copies stay bit-identical).  One host thread queues a
    step for all members without waiting; the minibatch worker uploads and gathers every member's share ahead
    (`ShardedMinibatchFeatures.prefetch_batch`) through an upload context per member."""

    def __init__(self, glm, feats, n_lik, children_by_member):
        _ResidentLoop.__init__(self, glm, feats, n_lik, children_by_member[0])
        self.group, self.kids = feats.group, children_by_member

    def begin(self, z0, lower, upper, updater, maxiter):
        _ResidentLoop.begin(self, z0, lower, upper, updater, maxiter)
        one = self._make   # (for the feature matrix `fm` of a member, made by `_start`)

        def make(M):
            sgds = []
            for i, f in enumerate(self.feats.feats):
                with _hip.device_scope(self.group.members[i]):
                    f._ensure(M, int(self.glm.D_))
                    sgds.append(one(M, fm=f.fm, children=self.kids[i]))
            return _hip.ResidentSgdGroup(self.group._comms, sgds)
        self._make_group = make

    def _start(self, M):
        # (every member's feature matrix holds a whole minibatch: which share of it falls to a member varies by step)
        self.sgd = self._make_group(M)

    def step(self, batch):
        g, feats = self.glm, self.feats
        batch = _gathered(batch)
        y, largs = batch[1], list(batch[2:])
        draws = spec = gathered = None
        while largs and isinstance(largs[-1], (_Draws, _Spec, _Batch)):      # made ahead on the worker (`_ahead`)
            last = largs.pop()
            if isinstance(last, _Draws):
                draws = last.e
            elif isinstance(last, _Batch):
                gathered = last.token
            else:
                spec = last.spec
        idx = largs.pop()
        if self.n_lik:
            from .likelihoods import RR_LIK_GAUSSIAN
            lid, rowarg, llconst = RR_LIK_GAUSSIAN, None, 0.0
        else:
            lid, _, rowarg, llconst = spec if spec is not None else g.likelihood.device_spec(y, [], largs)
        it = g._iteration(advance=1)
        dolog = (it % LOGITER == 0) or (it == g.maxiter - 1)
        if self.sgd is None:
            self._start(len(idx))
        shown = self._values() if dolog else None
        if gathered is not None and gathered.key != (id(y), len(y), rowarg is None):
            gathered = None
        parts = gathered.parts if gathered is not None else feats._split_idx(idx)
        mine = {i: (pos, local) for i, pos, local in parts}
        seed = key = 0
        if g.sampler == "device":
            seed, key = g._dev_seed, g._dev_step
            g._dev_step += 1
        elif draws is None:
            draws = g._reference_draws()
        step = []
        for i, f in enumerate(feats.feats):
            if i not in mine:
                step.append((None, 0, None, None, None))
                continue
            pos, local = mine[i]
            with _hip.device_scope(self.group.members[i]):
                got = gathered.got[i] if gathered is not None else None
                if got is not None:
                    dy, dn = got.dy, got.dn
                else:
                    dy = f._stage("y", np.asarray(y)[pos], np.float32)
                    dn = None if rowarg is None else f._stage("rowarg", np.asarray(rowarg)[pos], np.float32)
                dX = f.batch_rows(local, got)
                dE = None
                if g.sampler != "device":
                    dE = draws[i] if isinstance(draws, tuple) else f._stage("E", draws, np.float32)
            step.append((dX, len(local), dy, dn, dE))
        self.sgd.step(step, lid, llconst, g.B_, g.nsamples, seed, key)
        self.clock.append(time.perf_counter())
        if dolog:
            log.info("Iter {}: ELBO = {}, reg = {}, like_hypers = {}, basis_hypers = {}"
                     .format(it, -self.sgd.objective(self.t), shown[0], shown[1], shown[2]))
        self.t += 1

    def abort(self):
        sgd, self.sgd = self.sgd, None
        if sgd is not None:
            self.group.sync()   # (a member's step may still read what another member's loop owns: all idle before any goes)
            sgd.close()
        for f in self.feats.feats:
            f.__dict__.pop("PREFETCH_SLOTS", None)


class _FusedLoop(object):
    """`optimize.sgd`'s loop AND structured_sgd's random starts for SMALL minibatches (the reference's defaults: 10 rows, K = 10,
    50 samples, 3000 steps, 500 starts -- glm.py:120-124): rr_glm_svi runs many SGD steps inside ONE kernel launch (K
    cooperating workgroups; rr_svi.hip) and scores all candidates of the random starts in one more.  The host's part of a
    step is what must stay in the reference's order on ONE RandomState -- cutting the minibatch (the permutation stream) and,
    with sampler="host", drawing the step's normals -- both on the prefetch worker; row indices and draws go up in blocks of
    steps.  X, y, the per-row likelihood argument, z, the updater's state: resident.  `_elbo` is never called."""

    log_coordinates = None
    BLOCK_BYTES = 48 << 20    # draws uploaded per launch at most (two device buffers of this size in turn)
    BLOCK_STEPS = 256

    def __init__(self, glm, feats, n_lik, children, y, likelihood_args):
        from . import optimize as opt
        self.glm, self.feats, self.n_lik, self.children = glm, feats, n_lik, children
        self.y, self.largs = np.asarray(y, dtype=float), tuple(likelihood_args)
        self.svi = None
        self.updater = opt.Adam() if glm.updater is None else glm.updater
        self.M = int(min(glm.batch_size, len(self.y)))
        self._cands, self._start_idx, self._start_draws = [], [], []
        self._bufs = {}

    # -- the device object ---------------------------------------------------------------------------------------------------
    def _ensure(self):
        if self.svi is not None:
            return self.svi
        g, feats = self.glm, self.feats
        dev = _hip.get_device()
        self.dev = dev
        kind = type(self.updater).__name__
        u = self.updater
        par = {"SGDUpdater": lambda: [u.eta], "AdaDelta": lambda: [u.rho, u.epsilon], "AdaGrad": lambda: [u.eta, u.epsilon],
               "Momentum": lambda: [u.rho, u.eta], "Adam": lambda: [u.alpha, u.beta1, u.beta2, u.epsilon]}[kind]()
        self.dy = dev.upload_vector(self.y, np.float64)
        self.dn = dev.upload_vector(np.asarray(self.largs[0], dtype=float), np.float64) if self.largs and len(self.largs[0]) else None
        from .likelihoods import RR_LIK_GAUSSIAN
        if self.n_lik:
            lik = RR_LIK_GAUSSIAN
        else:
            lik = g.likelihood.device_spec(self.y[:1], [], [a[:1] for a in self.largs if len(a)])[0]
        # per row, the part of loglike that does not depend on f (likelihoods.py:171-192, 456-481: the log-factorials)
        from scipy.special import gammaln
        from .likelihoods import RR_LIK_BINOMIAL, RR_LIK_POISSON_EXP, RR_LIK_POISSON_SOFTPLUS
        lc = None
        if lik in (RR_LIK_POISSON_EXP, RR_LIK_POISSON_SOFTPLUS):
            lc = -gammaln(self.y + 1.0)
        elif lik == RR_LIK_BINOMIAL:
            n = np.asarray(self.largs[0], dtype=float)
            lc = gammaln(n + 1.0) - gammaln(self.y + 1.0) - gammaln(n - self.y + 1.0)
        self.dlc = dev.upload_vector(lc, np.float64) if lc is not None else None
        kids = [c + (kid.dX,) for c, kid in zip(self.children, feats._kids)]
        F = int(g.D_)
        self.np_ = 2 * F * g.K + len(kids) + self.n_lik + sum(c[2] for c in self.children if c[0] == "rff")
        hold = np.zeros(self.np_)
        self.svi = _hip.FusedSvi(dev, kids, len(self.y), self.dy, self.dn, self.dlc, g.K, g.nsamples, self.M, lik, self.n_lik, hold,
                                 np.full(self.np_, -np.inf), np.full(self.np_, np.inf), np.zeros(self.np_, dtype=np.uint8),
                                 _hip.UPDATER_IDS[kind], par, max(1, int(g.maxiter)), g.B_)
        self.F = F
        return self.svi

    def _device(self, name, nbytes):
        buf = self._bufs.get(name)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                self.dev.sync()
                buf.free()
            buf = self._bufs[name] = self.dev.malloc(max(int(nbytes), 4))
        return buf

    def _up(self, name, arr):
        arr = np.ascontiguousarray(arr)
        buf = self._device(name, arr.nbytes)
        if arr.nbytes:
            _hip._check(self.dev.lib, self.dev.lib.rr_memcpy_h2d(self.dev.ctx, buf.ptr, arr.ctypes.data_as(_hip.ctypes.c_void_p), arr.nbytes))
        return buf

    @staticmethod
    def _split(batch):
        """(row indices, draws or None) of a batch as the generator / the prefetch worker hands it over"""
        items = list(batch)
        draws = None
        while items and isinstance(items[-1], (_Draws, _Spec, _Batch)):
            last = items.pop()
            if isinstance(last, _Draws):
                draws = last.e
        return np.asarray(items[-1]), draws

    # -- the random starts (structured_sgd: decorators.py:541-583) -------------------------------------------------------------
    def note_start(self, batch, cand_flat):
        """One candidate: its minibatch (already cut), its parameter draw (already made); with the reference's stream its
        `_elbo`'s normals are drawn from `random_` NOW -- the order of a sequential run (batch, candidate, draws)."""
        g = self.glm
        idx, _ = self._split(batch)
        self._start_idx.append(np.asarray(idx, dtype=np.int32))
        self._cands.append(np.asarray(cand_flat, dtype=float))
        if g.sampler != "device":
            self._start_draws.append(g._reference_draws())

    def score_starts(self):
        """-ELBO of every noted candidate, in order: one launch."""
        g = self.glm
        svi = self._ensure()
        ns = len(self._cands)
        didx = self._up("sidx", np.stack(self._start_idx))
        if g.sampler == "device":
            objs = svi.starts(didx, np.stack(self._cands), None, g._dev_seed, g._dev_step)
            g._dev_step += ns
        else:
            objs = svi.starts(didx, np.stack(self._cands), self._up("sE", np.stack(self._start_draws)))
        g._iteration(advance=ns)
        self._cands, self._start_idx, self._start_draws = [], [], []
        self._bufs.pop("sE").free() if "sE" in self._bufs else None
        return objs

    # -- the loop (optimize.sgd: device_loop protocol) -----------------------------------------------------------------------
    def begin(self, z0, lower, upper, updater, maxiter):
        if not np.isfinite(maxiter):
            raise ValueError("the fused loop needs a finite maxiter")
        if updater is not None and (type(updater) is not type(self.updater) or vars(updater) != vars(self.updater)):
            raise ValueError("the fused loop was built for another updater")
        svi = self._ensure()
        pos = self.log_coordinates if self.log_coordinates is not None else np.zeros(len(z0), dtype=bool)
        self.pos = np.asarray(pos, dtype=bool)
        svi.set_start(z0, lower, upper, self.pos)
        g = self.glm
        per_step = g.K * g.nsamples * self.F * 4 if g.sampler != "device" else 0
        self.T = int(max(1, min(self.BLOCK_STEPS, self.BLOCK_BYTES // max(per_step, 1), max(1, int(maxiter)))))
        self._idx = np.empty((self.T, self.M), dtype=np.int32)
        self._E = [np.empty((self.T, g.K * g.nsamples, self.F), dtype=np.float32) for _ in range(2)] if per_step else None
        self._fill, self._turn, self.t = 0, 0, 0
        self._z0 = np.array(z0, dtype=float)
        self.clock = []

    def _values(self, z):
        x = np.where(self.pos, np.exp(np.where(self.pos, z, 0.0)), z)
        o, nk = 2 * self.glm.D_ * self.glm.K, len(self.children)
        regs = list(x[o:o + nk])
        ls, q = [], o + nk + self.n_lik
        for c in self.children:
            n = c[2] if c[0] == "rff" else 0
            if n:
                ls.append(x[q] if n == 1 else x[q:q + n])
            q += n
        return (regs[0] if nk == 1 else regs), ([x[o + nk]] if self.n_lik else []), (ls[0] if len(ls) == 1 else ls)

    def _flush(self):
        n = self._fill
        if n == 0:
            return
        g, svi = self.glm, self.svi
        # (two device buffers in turn: the launch that read buffer `turn` two flushes ago is over once the copy into it may
        # start -- rr_memcpy_h2d is ordered on the stream the launches run on)
        didx = self._up("idx%d" % self._turn, self._idx[:n])
        if g.sampler == "device":
            svi.run(n, didx, None, g._dev_seed, g._dev_step)
            g._dev_step += n
        else:
            svi.run(n, didx, self._up("E%d" % self._turn, self._E[self._turn][:n]))
        self._turn ^= 1
        self._fill = 0
        self.clock.append((time.perf_counter(), n))

    def step(self, batch):
        g = self.glm
        idx, draws = self._split(batch)
        it = g._iteration(advance=1)
        dolog = ((it % LOGITER == 0) or (it == g.maxiter - 1)) and log.isEnabledFor(logging.INFO)
        shown = None
        if dolog:   # the log line shows the parameters this step STARTS from: everything before it runs first
            self._flush()
            shown = self._values(self.svi.read()[0])
        self._idx[self._fill] = idx
        if g.sampler != "device":
            self._E[self._turn][self._fill] = draws if draws is not None else g._reference_draws()
        self._fill += 1
        self.t += 1
        if self._fill == self.T or dolog:
            self._flush()
        if dolog:
            log.info("Iter {}: ELBO = {}, reg = {}, like_hypers = {}, basis_hypers = {}"
                     .format(it, -self.svi.read()[1][self.t - 1], shown[0], shown[1], shown[2]))

    def end(self):
        if self.svi is None or self.t == 0:
            z0 = getattr(self, "_z0", None)
            self.abort()
            return z0, np.empty(0), np.empty(0)
        self._flush()
        z, objs, norms = self.svi.read()
        self.clock.append((time.perf_counter(), 0))
        self.glm.__dict__["_resident_clock"] = np.array([c[0] for c in self.clock])
        self.abort()
        return z, objs, norms

    def abort(self):
        svi, self.svi = self.svi, None
        if svi is not None:
            svi.close()
        for b in self._bufs.values():
            b.free()
        self._bufs = {}
        for name in ("dy", "dn", "dlc"):
            b = self.__dict__.pop(name, None)
            if b is not None:
                b.free()


class _Batch(object):
    """A minibatch gathered on the device ahead of its step, riding behind the batch's arguments (`_ahead`)."""

    def __init__(self, token):
        self.token = token


class _Spec(object):
    """`likelihood.device_spec` of one minibatch, evaluated ahead of its step."""

    def __init__(self, spec):
        self.spec = spec


class _Draws(object):
    """Standard-normal draws of one SVI step, riding along with its minibatch."""

    def __init__(self, e):
        self.e = e


def _flat_params(p):
    """the Parameter objects of a (nested) parameter list, in order"""
    if isinstance(p, (list, tuple)) and not isinstance(p, Parameter):
        return [q for item in p for q in _flat_params(item)]
    return [p]


def _like_structure(template, flat):
    """`flat` cut back into the nesting of `template` (scalars, arrays, lists of them, [])."""
    pos = [0]

    def build(t):
        if isinstance(t, list):
            return [build(u) for u in t]
        n = int(np.size(t))
        v = flat[pos[0]:pos[0] + n]
        pos[0] += n
        return float(v[0]) if np.ndim(t) == 0 else np.array(v).reshape(np.shape(t))
    return build(template)


def _reshape_likelihood_args(likelihood_args, N):
    """Per-observation likelihood arguments as length-N vectors: scalars are broadcast, empty sequences pass."""
    out = []
    for arg in likelihood_args:
        vec = np.full(N, arg, dtype=float) if np.isscalar(arg) else arg
        if len(vec) not in (0, N):
            raise ValueError("Likelihood arguments not a compatible shape!")
        out.append(vec)
    return tuple(out)


def _bisect_quantile(cdf, p, reach, iters=100):
    """q with cdf(q) = p for every row, cdf monotone in q and evaluated for all rows at once: bisection on
    [-reach, reach]; NaN for rows whose bracket does not straddle p."""
    lo, hi = -reach, reach.copy()
    with np.errstate(invalid="ignore"):
        inside = (cdf(lo) <= p) & (cdf(hi) >= p)
        for _ in range(iters):
            mid = 0.5 * (lo + hi)
            below = cdf(mid) < p
            lo = np.where(below, mid, lo)
            hi = np.where(below, hi, mid)
    return np.where(inside, 0.5 * (lo + hi), np.nan)


def _qmatrix(m, C):
    """logq[j, i] = log N(m_i | m_j, diag(C_i + C_j))  (glm.py:697-712), vectorised."""
    D = m.shape[0]
    dc = C[:, :, np.newaxis] + C[:, np.newaxis, :]                  # D x i x j
    dm2 = (m[:, :, np.newaxis] - m[:, np.newaxis, :]) ** 2
    return (-0.5 * (D * np.log(2 * np.pi) + np.log(dc).sum(axis=0) + (dm2 / dc).sum(axis=0))).T
