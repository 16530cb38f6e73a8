"""
Optimiser front-ends the estimators use: nested ``Parameter`` structures in, scipy's
L-BFGS-B underneath, with the log trick for ``Positive`` bounds.  Host control plane that
runs once per iteration (reference: revrand/optimize/decorators.py:24-130, 255-326,
541-617), restated compactly rather than as a decorator stack.
"""
import logging

import numpy as np

from .btypes import Bound, Positive
from .utils import flatten_values, issequence, shapes_of, unflatten

log = logging.getLogger(__name__)

MINPOS = 1e-100                             # decorators.py:18
MAXPOS = np.sqrt(np.finfo(float).max)       # decorators.py:19
LOGMINPOS, EXPMAX = np.log(MINPOS), np.log(MAXPOS)


def _map(fn, params):
    return [_map(fn, p) for p in params] if issequence(params) else fn(params)


def _flat_bounds(params):
    out = []

    def walk(p):
        if issequence(p):
            for q in p:
                walk(q)
        else:
            out.extend([p.bounds] * int(np.prod(p.shape, dtype=int)))
    walk(params)
    return out


def _random_start(fun, parameters, jac, args, nstarts, random_state):
    """Best of `nstarts` draws from the Parameters' distributions (decorators.py:541-583)."""
    if nstarts < 1:
        raise ValueError("nstarts has to be greater than or equal to 1")
    if not any(flatten_values(_map(lambda p: float(p.is_random), parameters))):
        log.info("No random parameters, not doing any random starts")
        return flatten_values(_map(lambda p: p.value, parameters))
    log.info("Evaluating random starts...")
    best_obj, best = np.inf, None
    # a function may offer a cheaper objective-only evaluation (the starts are ranked by the objective alone)
    objective_only = getattr(fun, "objective_only", None)
    for _ in range(nstarts):
        cand = _map(lambda p: p.rvs(random_state), parameters)
        if objective_only is not None:
            obj = objective_only(*(list(cand) + list(args)))
        else:
            res = fun(*(list(cand) + list(args)))
            obj = res[0] if jac is True else res
        if best is None or obj < best_obj:
            best_obj, best = obj, cand
    log.info("Best start found with objective = {}".format(best_obj))
    return flatten_values(best)


def structured_minimizer(minimizer):
    """Let `minimizer` (scipy.optimize.minimize-like) take nested Parameter lists.

    new_minimizer(fun, parameters, jac=True, args=(), nstarts=0, random_state=None, **kw):
    `fun(*values)` receives one value per top-level Parameter (python float for scalars,
    ndarray otherwise, [] for the null Parameter, nested lists kept) and returns
    ``(objective, nested_gradients)`` when jac is True.  ``result.x`` / ``result.jac`` are
    returned in the same nested structure (decorators.py:24-130).
    """
    def new_minimizer(fun, parameters, jac=True, args=(), nstarts=0, random_state=None, start_values=None, **kwargs):
        shapes = shapes_of(parameters, shape=lambda p: p.shape)
        if start_values is not None:   # (not in the reference: restart from given values, no draws -- StandardLinearModel's retry)
            x0 = flatten_values(start_values)
        else:
            x0 = flatten_values(_map(lambda p: p.rvs(random_state), parameters))
        bounds = _flat_bounds(parameters)
        if nstarts > 0 and start_values is None:
            x0 = _random_start(fun, parameters, jac, args, nstarts, random_state)

        def flat_fun(x, *fargs):
            res = fun(*(unflatten(x, shapes) + list(fargs)))
            if jac is True:
                return res[0], flatten_values(res[1])
            return res

        flat_jac = jac
        if callable(jac):
            def flat_jac(x, *fargs):
                return flatten_values(jac(*(unflatten(x, shapes) + list(fargs))))

        result = minimizer(flat_fun, x0, jac=flat_jac, args=args, bounds=bounds, **kwargs)
        result["x"] = tuple(unflatten(result["x"], shapes))
        if bool(jac) and "jac" in result:
            result["jac"] = tuple(unflatten(result["jac"], shapes))
        return result
    return new_minimizer


def logtrick_minimizer(minimizer):
    """Optimise log(x) for every variable with a ``Positive`` bound (decorators.py:255-326)."""
    def new_minimizer(fun, x0, jac=True, bounds=None, **kwargs):
        if bounds is None:
            return minimizer(fun, x0, jac=jac, bounds=bounds, **kwargs)
        pos = np.array([isinstance(b, Positive) for b in bounds], dtype=bool)

        def to_log(x):
            z = np.array(x, dtype=float)
            z[pos] = np.log(z[pos])
            return z

        def from_log(z):
            x = np.array(z, dtype=float)
            x[pos] = np.exp(x[pos])
            return x

        def chain(g, z):
            g = np.array(g, dtype=float)
            g[pos] *= np.exp(z[pos])
            return g

        new_bounds = [Bound(LOGMINPOS, EXPMAX if b.upper is None else np.log(b.upper)) if p else b
                      for b, p in zip(bounds, pos)]

        if callable(jac):
            def new_jac(z, *a, **k):
                return chain(jac(from_log(z), *a, **k), z)
        else:
            new_jac = jac

        if (not callable(jac)) and bool(jac):
            def new_fun(z, *a, **k):
                o, g = fun(from_log(z), *a, **k)
                return o, chain(g, z)
        else:
            def new_fun(z, *a, **k):
                return fun(from_log(z), *a, **k)

        result = minimizer(new_fun, to_log(x0), jac=new_jac, bounds=new_bounds, **kwargs)
        result["x"] = from_log(result["x"])
        return result
    return new_minimizer


# --------------------------------------------------------------------------------------
# Stochastic gradients (reference: optimize/sgd.py, optimize/decorators.py:133-252,329-408)
# --------------------------------------------------------------------------------------

class SGDUpdater(object):
    """x - eta * grad (sgd.py:14-68)."""

    def __init__(self, eta=0.1):
        self.eta = eta

    def __call__(self, x, grad):
        return x - self.eta * grad

    def reset(self):
        pass

    def __repr__(self):
        return "{}(eta={})".format(type(self).__name__, self.eta)


class AdaDelta(SGDUpdater):
    """sgd.py:71-133."""

    def __init__(self, rho=0.1, epsilon=1e-5):
        if rho < 0 or rho > 1:
            raise ValueError("Decay rate 'rho' must be between 0 and 1!")
        if epsilon <= 0:
            raise ValueError("Constant 'epsilon' must be > 0!")
        self.rho, self.epsilon = rho, epsilon
        self.Eg2 = self.Edx2 = 0

    def __call__(self, x, grad):
        self.Eg2 = self.rho * self.Eg2 + (1 - self.rho) * grad ** 2
        dx = -grad * np.sqrt(self.Edx2 + self.epsilon) / np.sqrt(self.Eg2 + self.epsilon)
        self.Edx2 = self.rho * self.Edx2 + (1 - self.rho) * dx ** 2
        return x + dx

    def reset(self):
        self.__init__(self.rho, self.epsilon)

    def __repr__(self):
        return "{}(rho={}, epsilon={})".format(type(self).__name__, self.rho, self.epsilon)


class AdaGrad(SGDUpdater):
    """sgd.py:136-196."""

    def __init__(self, eta=1, epsilon=1e-6):
        if eta <= 0:
            raise ValueError("Learning rate 'eta' must be > 0!")
        if epsilon <= 0:
            raise ValueError("Constant 'epsilon' must be > 0!")
        self.eta, self.epsilon = eta, epsilon
        self.g2_hist = 0

    def __call__(self, x, grad):
        self.g2_hist = self.g2_hist + grad ** 2
        return x - self.eta * grad / (self.epsilon + np.sqrt(self.g2_hist))

    def reset(self):
        self.__init__(self.eta, self.epsilon)

    def __repr__(self):
        return "{}(eta={}, epsilon={})".format(type(self).__name__, self.eta, self.epsilon)


class Momentum(SGDUpdater):
    """sgd.py:199-256."""

    def __init__(self, rho=0.5, eta=0.01):
        if eta <= 0:
            raise ValueError("Learning rate 'eta' must be > 0!")
        if rho < 0 or rho > 1:
            raise ValueError("Decay rate 'rho' must be between 0 and 1!")
        self.eta, self.rho = eta, rho
        self.dx = 0

    def __call__(self, x, grad):
        self.dx = self.rho * self.dx - self.eta * grad
        return x + self.dx

    def reset(self):
        self.__init__(self.rho, self.eta)

    def __repr__(self):
        return "{}(rho={}, eta={})".format(type(self).__name__, self.rho, self.eta)


class Adam(SGDUpdater):
    """sgd.py:259-330."""

    def __init__(self, alpha=0.01, beta1=0.9, beta2=0.99, epsilon=1e-8):
        self.alpha, self.beta1, self.beta2, self.epsilon = alpha, beta1, beta2, epsilon
        self.t = 0
        self.m = self.v = None

    def __call__(self, x, grad):
        self.t += 1
        if self.m is None:
            self.m = np.zeros_like(x)
            self.v = np.zeros_like(x)
        self.m = self.beta1 * self.m + (1 - self.beta1) * grad
        self.v = self.beta2 * self.v + (1 - self.beta2) * grad ** 2
        mbar = self.m / (1 - self.beta1 ** self.t)
        vbar = self.v / (1 - self.beta2 ** self.t)
        return x - self.alpha * mbar / (np.sqrt(vbar) + self.epsilon)

    def reset(self):
        self.__init__(self.alpha, self.beta1, self.beta2, self.epsilon)

    def __repr__(self):
        return "{}(alpha={}, beta1={}, beta2={}, epsilon={})".format(type(self).__name__, self.alpha, self.beta1,
                                                                     self.beta2, self.epsilon)


def _len_data(data):
    if not issequence(data):
        return data.shape[0]
    N = len(data[0])
    for d in data[1:]:
        if d.shape[0] != N:
            raise ValueError("Not all data is the same length!")
    return N


def _legacy_permutation(generator, N, out=None):
    if N < 4096 or not isinstance(generator, np.random.RandomState):
        return generator.permutation(N)
    try:  # (the generic host optimisers must keep working where the native library is not built or loadable)
        from . import _hip
        return _hip.legacy_permutation(generator, N, out=out)
    except (ImportError, OSError, AttributeError):
        perm = generator.permutation(N)
        if out is not None:
            out[:] = perm
            return out
        return perm


def gen_batch(data, batch_size, maxiter=np.inf, random_state=None, _reuse=False):
    """Minibatches by sweeping random permutations of the rows (sgd.py:428-470).

    The index stream is exactly ``endless_permutations`` (utils/rand.py:7-31) -- a new ``permutation(N)`` is drawn
    from `random_state` only at the moment the previous one is used up, so the draws interleave with the caller's
    other uses of the same RandomState as in the reference -- but a batch is cut from the permutation arrays
    instead of 65 536 ``next()`` calls per step."""
    from sklearn.utils import check_random_state
    generator = check_random_state(random_state)
    N = _len_data(data)
    perm, pos = np.empty(0, dtype=int), 0
    it = 0
    # Long epochs (N >= 16 batches): the permutations go into three arrays in turn instead of a fresh 8 N bytes per epoch -- a
    # batch is a VIEW of its permutation, and fewer than 16 batches are ever alive at once (the prefetch pipeline's queues).
    # The allocator maps and unmaps arrays of that size; in a process with hundreds of threads the page faults and TLB
    # shootdowns of that stalled every thread of config 5's fit for 15-30 ms at each epoch boundary.
    # (`_reuse`: only for a consumer that holds fewer than 16 batches at a time -- `sgd`; anybody else may keep every batch)
    ring = [np.empty(N, dtype=np.int64) for _ in range(3)] if (_reuse and N >= 4096 and N >= 16 * batch_size) else None
    for r in ring or ():
        r.fill(0)  # (touched now: the first use of an untouched 16 MB array is 4096 page faults in the middle of the loop)
    turn = 0
    while it < maxiter:
        it += 1
        parts, need = [], batch_size
        while need > 0:
            if pos == len(perm):
                # (the library's restatement of RandomState.permutation: same values, same state, but outside the GIL -- NumPy's
                # loop holds it for 18 ms at N = 2M, and the thread consuming the batches stands still with it)
                perm, pos = _legacy_permutation(generator, N, None if ring is None else ring[turn % 3]), 0
                turn += 1
            take = min(need, len(perm) - pos)
            parts.append(perm[pos:pos + take])
            pos += take
            need -= take
        ind = parts[0] if len(parts) == 1 else np.concatenate(parts)
        yield (data[ind],) if not issequence(data) else [d[ind] for d in data]


def _prefetched(batches, augment=None):
    """The same batches, each produced one step ahead on a worker thread: the device calls inside `fun` release the
    GIL, so the permutation draw and the row gather of step t+1 overlap the kernels of step t.  The order of the
    batches, hence the generator's use of its RandomState, is unchanged.  `augment(batch) -> batch` runs on the worker
    right after a batch is cut (the GLM draws its step's standard normals there: batch_t, e_t, batch_t+1, e_t+1, ... is
    exactly the order in which a sequential run consumes the stream).  A LIST of callables is a pipeline: the first runs on
    the thread that cuts the batches, every further one on a thread of its own behind a one-batch queue (the GLM: draws on
    the first -- they share the RandomState with the permutations -- uploads and gathers on the second; the slower of the
    two, not their sum, bounds the rate at which batches arrive)."""
    import queue
    import threading
    stages = [f for f in (augment if isinstance(augment, (list, tuple)) else [augment]) if f is not None] or [None]
    # (one finished batch waits behind every stage; behind the FIRST stage of a pipeline, eight: what it makes lives in host
    # memory, and the thread that cuts the batches stalls for a whole permutation(N) at every epoch boundary -- 18 ms at
    # N = 2M, five steps' worth at config 5 -- which the later stages and the consumer then do not notice)
    qs = [queue.Queue(maxsize=8 if (i == 0 and len(stages) > 1) else 1) for i in range(len(stages))]
    stop, END = threading.Event(), object()

    def put(q, item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def get(q):
        while not stop.is_set():
            try:
                return q.get(timeout=0.1)
            except queue.Empty:
                pass
        return END

    def first():
        try:
            for item in batches:
                if stages[0] is not None:
                    item = stages[0](item)
                if not put(qs[0], item):
                    return
            put(qs[0], END)
        except BaseException as e:  # surfaces in the consumer
            put(qs[0], e)

    def later(i):
        try:
            while True:
                item = get(qs[i - 1])
                if item is END or isinstance(item, BaseException):
                    put(qs[i], item)
                    return
                if not put(qs[i], stages[i](item)):
                    return
        except BaseException as e:
            put(qs[i], e)

    workers = [threading.Thread(target=first, daemon=True)] + \
              [threading.Thread(target=later, args=(i,), daemon=True) for i in range(1, len(stages))]
    for w in workers:
        w.start()
    try:
        while True:
            item = qs[-1].get()
            if item is END:
                return
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        # a worker may be in the middle of a batch (`augment` uploads and gathers on the device for the GLM): let it finish
        # that one before the caller frees what it writes to.  (`sgd` closes this generator explicitly when its objective
        # raises: the traceback would otherwise keep it -- and the workers -- alive past the caller's cleanup.)
        for w in workers:
            w.join(timeout=60.0)
            if w.is_alive():
                log.error("a minibatch prefetch worker is still running 60 s after it was told to stop: device buffers it "
                          "writes to may be freed under it")


def sgd(fun, x0, data, args=(), bounds=None, batch_size=10, maxiter=5000, updater=None, eval_obj=False,
        random_state=None, prefetch=False, device_loop=None):
    """Stochastic gradient descent over minibatches of `data` (sgd.py:337-425): ``fun(x, *batch, *args)`` returns
    the gradient (or ``(objective, gradient)`` with eval_obj); bounded coordinates have outward gradients
    truncated and steps clipped.  `prefetch` (not in the reference): build each minibatch one step ahead on a worker
    thread -- only valid when `fun` does not draw from `random_state` itself; a callable is applied to every batch on
    that thread (``_prefetched``).  `device_loop` (not in the reference): an object that runs THIS loop with x, the updater's
    state and the gradient in device memory -- ``begin(x0, lower, upper, updater, maxiter)``, ``step(batch + args)`` per
    minibatch (queues the step, returns at once), ``end() -> (x, objs, norms)``, ``abort()``; `fun` is not called then
    (GeneralizedLinearModel's resident SVI loop: rr_glm_sgd, glm._ResidentLoop)."""
    from scipy.optimize import OptimizeResult
    if updater is None:
        updater = Adam()
    updater.reset()
    N = _len_data(data)
    x = np.array(x0, copy=True, dtype=float)
    batch_size = min(batch_size, N)
    if bounds is not None:
        if len(bounds) != x.shape[0]:
            raise ValueError("The dimension of the bounds does not match x0!")
        lower = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=float)
        upper = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=float)
    obj, objs, norms = None, [], []
    batches = gen_batch(data, batch_size, maxiter, random_state, _reuse=True)
    ahead = _prefetched(batches, prefetch if (callable(prefetch) or isinstance(prefetch, (list, tuple))) else None) if prefetch else None
    if device_loop is not None:
        if bounds is None:
            lower, upper = np.full(x.shape, -np.inf), np.full(x.shape, np.inf)
        try:
            device_loop.begin(x, lower, upper, updater, maxiter)
            for batch in (ahead if ahead is not None else batches):
                device_loop.step(list(batch) + list(args))
            x, objs, norms = device_loop.end()
        except BaseException:
            device_loop.abort()
            raise
        finally:
            if ahead is not None:
                ahead.close()
        objs, norms = [float(o) for o in objs], [float(v) for v in norms]
        return OptimizeResult(x=x, norms=norms, message='maxiter reached', fun=objs[-1] if objs else None, objs=objs)
    try:
        for batch in (ahead if ahead is not None else batches):
            if not eval_obj:
                grad = fun(x, *(list(batch) + list(args)))
            else:
                obj, grad = fun(x, *(list(batch) + list(args)))
                objs.append(obj)
            norms.append(float(np.sqrt(np.square(grad).sum())))  # np.linalg.norm, without a threaded BLAS call per step
            if bounds is not None:
                xlower = x <= lower
                if xlower.any():  # (rarely: a coordinate sitting ON its bound; the masked gather / scatter only then)
                    grad[xlower] = np.minimum(grad[xlower], 0)
                xupper = x >= upper
                if xupper.any():
                    grad[xupper] = np.maximum(grad[xupper], 0)
            x = updater(x, grad)
            if bounds is not None:
                x = np.clip(x, lower, upper)
    finally:
        if ahead is not None:
            ahead.close()  # stops and joins the worker NOW: the caller frees what the worker's batches write to next
    return OptimizeResult(x=x, norms=norms, message='maxiter reached', fun=obj, objs=objs)


def structured_sgd(sgd):
    """Let `sgd` take nested Parameter lists (decorators.py:133-252).

    new_sgd(fun, parameters, data, eval_obj=False, batch_size=10, args=(), random_state=None, nstarts=100, **kw):
    ``fun(*values, *batch, *args)``.  With eval_obj and nstarts > 0 the best of `nstarts` random draws of the
    Parameters (each scored on its own minibatch) is the starting point.  Unlike the reference -- which drops
    ``batch_size`` here so its main loop always runs sgd's default of 10 rows (decorators.py:244-246) --
    ``batch_size`` is forwarded to `sgd`; the two agree at the reference's default batch_size=10.
    """
    def new_sgd(fun, parameters, data, eval_obj=False, batch_size=10, args=(), random_state=None, nstarts=100,
                sync=None, **sgd_kwargs):
        shapes = shapes_of(parameters, shape=lambda p: p.shape)
        nparams = len(shapes)
        # the start point is a DRAW of the random Parameters, as in the reference (its `flatten` ravels
        # `parameter.rvs(random_state=None)`, btypes.py:351-371, decorators.py:216-220): NumPy's global stream, which
        # leaves `random_state` -- the minibatch / reparameterisation stream -- exactly where the reference has it.
        # (With all K mixture components at the distribution mean the GLM's mixture would start out degenerate.)
        x0 = flatten_values(_map(lambda p: p.rvs(np.random.mtrand._rand), parameters))
        bounds = _flat_bounds(parameters)
        # `sync` (row-sharded fits): every candidate and the start point are rank 0's -- the ranks' streams differ as
        # soon as their shard sizes do (a permutation of N_local rows consumes N_local-dependent state)
        share = (lambda v: np.asarray(sync(np.ascontiguousarray(v, dtype=float)))) if sync is not None else (lambda v: v)
        x0 = share(x0)
        if eval_obj and nstarts > 0:
            data_gen = gen_batch(data, batch_size, random_state=random_state)
            if any(flatten_values(_map(lambda p: float(p.is_random), parameters))):
                log.info("Evaluating random starts...")
                best_obj, best = np.inf, None
                objective_only = getattr(fun, "objective_only", None)
                # a device loop may score ALL candidates in one launch (glm._FusedLoop): the candidates, their minibatches
                # and -- with the reference's random stream -- their draws are made here in the reference's order
                # (batch, candidate, the evaluation's draws: decorators.py:560-566), the evaluations happen afterwards
                batched = sgd_kwargs.get("device_loop") if sync is None else None
                if batched is not None and getattr(batched, "note_start", None) is not None:
                    cands = []
                    for _ in range(nstarts):
                        batch = next(data_gen)
                        cand = _map(lambda p: p.rvs(random_state), parameters)
                        batched.note_start(list(batch) + list(args), flatten_values(cand))
                        cands.append(cand)
                    for cand, obj in zip(cands, batched.score_starts()):
                        if best is None or obj < best_obj:
                            best_obj, best = obj, cand
                    nstarts = 0
                for _ in range(nstarts):
                    batch = next(data_gen)
                    cand = _map(lambda p: p.rvs(random_state), parameters)
                    if sync is not None:
                        cand = unflatten(share(flatten_values(cand)), shapes)[:nparams]
                    if objective_only is not None:  # a cheaper objective-only evaluation, when the function offers one
                        obj = objective_only(*(list(cand) + list(batch) + list(args)))
                    else:
                        obj = fun(*(list(cand) + list(batch) + list(args)))[0]
                    if best is None or obj < best_obj:
                        best_obj, best = obj, cand
                log.info("Best start found with objective = {}".format(best_obj))
                x0 = flatten_values(best)
            else:
                log.info("No random parameters, not doing any random starts")

        def flat_fun(x, *rest):
            res = fun(*(unflatten(x, shapes)[:nparams] + list(rest)))
            if eval_obj:
                return res[0], flatten_values(res[1])
            return flatten_values(res)

        result = sgd(flat_fun, x0, data=data, bounds=bounds, args=args, eval_obj=eval_obj, batch_size=batch_size,
                     random_state=random_state, **sgd_kwargs)
        result["x"] = tuple(unflatten(result["x"], shapes))
        return result
    return new_sgd


def logtrick_sgd(sgd):
    """Optimise log(x) for every variable with a ``Positive`` bound (decorators.py:329-408)."""
    def new_sgd(fun, x0, data, bounds=None, eval_obj=False, **sgd_kwargs):
        if bounds is None:
            return sgd(fun, x0, data, bounds=bounds, eval_obj=eval_obj, **sgd_kwargs)
        pos = np.array([isinstance(b, Positive) for b in bounds], dtype=bool)
        if sgd_kwargs.get("device_loop") is not None:  # the device loop applies from_log / the chain rule itself
            sgd_kwargs["device_loop"].log_coordinates = pos
        # the Positive coordinates as contiguous RUNS (a (D, K) covariance block is one run of D K entries): exp over slices
        # instead of boolean-mask gathers and scatters of the whole vector -- same values, 0.25 ms less host time per SGD
        # step at config 5's 41 000 coordinates, where the step's serial host part is what the GPU waits for
        edges = np.flatnonzero(np.diff(np.concatenate(([0], pos.astype(np.int8), [0]))))
        runs = list(zip(edges[0::2].tolist(), edges[1::2].tolist()))

        def from_log(z):
            x = np.array(z, dtype=float)
            for a, b in runs:
                np.exp(x[a:b], out=x[a:b])
            return x

        def chain(g, z, x=None):
            """dx/dz = exp(z) = x on the Positive runs: `x` is from_log(z) of the SAME evaluation (one exp pass per step, not two)."""
            g = np.array(g, dtype=float)
            for a, b in runs:
                g[a:b] *= x[a:b] if x is not None else np.exp(z[a:b])
            return g

        new_bounds = [Bound(LOGMINPOS, EXPMAX if b.upper is None else np.log(b.upper)) if p else b
                      for b, p in zip(bounds, pos)]
        if eval_obj:
            def new_fun(z, *a, **k):
                x = from_log(z)
                o, g = fun(x, *a, **k)
                return o, chain(g, z, x)
        else:
            def new_fun(z, *a, **k):
                x = from_log(z)
                return chain(fun(x, *a, **k), z, x)
        z0 = np.array(x0, dtype=float)
        z0[pos] = np.log(z0[pos])
        result = sgd(new_fun, z0, data, bounds=new_bounds, eval_obj=eval_obj, **sgd_kwargs)
        result["x"] = from_log(result["x"])
        return result
    return new_sgd
