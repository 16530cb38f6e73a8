"""
Several GPUs behind ONE call in ONE process -- ``StandardLinearModel(basis, devices=[0, 1, ..., 7]).fit(X, y)``.

The reference's ``fit`` is a single call in a single process (revrand/slm.py:74-140) and its users drive it through
sklearn ``Pipeline`` / ``GridSearchCV`` objects (tests/test_models.py:39-80,150-238) that cannot be wrapped in a
per-GPU launcher.  ``revrand_amd.parallel`` covers the SPMD case (one process per GPU, RCCL between them); this module
covers the estimator's own call:

* a ``DeviceGroup`` owns one device context (``rr_ctx``: stream, scratch) per member and ONE host thread per member.
  Everything the single-device classes do -- handles, resident X / y, feature matrices, the posterior, the second pass --
  runs unchanged on a member's thread, because inside it ``_hip.get_device()`` resolves to the member's context
  (``_hip.device_scope``) and the per-basis handle caches are keyed by context (``_hip.device_key``);
* rows shard contiguously over the members (``parallel.shard_bounds``, as between ranks); every member accumulates the
  statistics ``[G | b | y^T y]`` of ITS rows on its own stream, then ONE in-process collective
  (``rr_comm_group_reduce_stats_dev``: pack the upper triangle -> all-reduce -> unpack + mirror, all stream-ordered) leaves
  the summed statistics in every member's HBM -- the same message as between ranks (``[tri G | b | y^T y | N]``, SURVEY
  8e), over RCCL (``ncclCommInitAll`` + group calls) when every member has its own GPU, or over the library's own peer
  transport (kernels loading the other members' buffers across the xGMI mesh; also what members sharing one GPU use);
* the posterior (``rr_posterior_dev``: Cholesky, inverse, reductions) is replicated -- every member forms it from its copy of
  the statistics, concurrently, so nothing F x F ever moves again -- and the second pass runs on every member's rows with the
  member's own copy of C; its ``1 + d`` numbers ``[sqErr | dhyp]`` come back to the host and are added there in member order.

Members that share a device (``devices=[0, 0]``: what a 1-GPU box can run) take exactly the same code path.
"""
import atexit
import ctypes
import logging
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _hip
from .parallel import shard_bounds, stats_count

log = logging.getLogger(__name__)

TRANSPORTS = {"auto": 0, "rccl": 1, "peer": 2}
_TRANSPORT_NAMES = {v: k for k, v in TRANSPORTS.items()}
_OPS = {"sum": 0, "max": 1, "min": 2}


def visible_devices():
    lib = _hip.load_library()
    n = ctypes.c_int()
    _hip._check(lib, lib.rr_device_count(ctypes.byref(n)))
    return n.value


def resolve_devices(devices):
    """``devices`` as the estimators take it -> a tuple of GPU indices: a sequence of indices (repeats allowed: two
    members on one GPU), an int n (GPUs 0 .. n-1) or "all" (every visible GPU)."""
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError("devices must be a sequence of GPU indices, a count or 'all'")
        devices = visible_devices()
    if np.isscalar(devices):
        if int(devices) < 1:
            raise ValueError("devices: at least one GPU")
        devices = range(int(devices))
    devices = tuple(int(d) for d in devices)
    if not devices:
        raise ValueError("devices: at least one GPU")
    return devices


class DeviceGroup(object):
    """n device contexts of this process, one host thread each, and the in-process communicator that sums their buffers.

    ``map(fn)`` runs ``fn(i)`` on member i's thread (all members concurrently; ctypes releases the GIL during a library
    call) and returns the results in member order; inside, ``_hip.get_device()`` is member i's context.
    """

    def __init__(self, devices, transport=None):
        self.devices = resolve_devices(devices)
        self.n = len(self.devices)
        self.pid = os.getpid()
        self.lib = _hip.load_library()
        transport = transport or os.environ.get("RR_COMM_TRANSPORT") or "auto"
        if transport not in TRANSPORTS:
            raise ValueError("transport must be one of %s" % sorted(TRANSPORTS))
        # contexts of their OWN (never the process-wide default context of a GPU: two members may share a GPU, and a member's
        # stream must not be the one an unrelated estimator of this process queues work on)
        self.members = [_hip.Device(d) for d in self.devices]
        wants_rccl = transport == "rccl" or os.environ.get("RR_COMM_TRANSPORT") == "rccl"   # (also a one-member group then)
        if transport != "peer" and len(set(self.devices)) == self.n and (self.n > 1 or wants_rccl):
            try:
                from .parallel import RcclComm
                RcclComm.load()  # the librccl paired with this process's HIP runtime, before ncclCommInitAll binds one
            except Exception as e:  # no loadable RCCL: the peer transport needs none
                if transport == "rccl":
                    raise
                log.info("RCCL not loadable (%s): the device group uses the peer transport", e)
                transport = "peer"
        ctxs = (ctypes.c_void_p * self.n)(*[m.ctx for m in self.members])
        comms = (ctypes.c_void_p * self.n)()
        _hip._check(self.lib, self.lib.rr_comm_init_all(self.n, ctxs, TRANSPORTS[transport], comms))
        self._comms = comms
        self.transport = _TRANSPORT_NAMES[self.lib.rr_comm_transport(comms[0])]
        self._msg = [None] * self.n
        self._pool = [ThreadPoolExecutor(max_workers=1, thread_name_prefix="rr-gpu%d-m%d" % (d, i),
                                         initializer=self._bind, initargs=(m,))
                      for i, (d, m) in enumerate(zip(self.devices, self.members))]
        self._lock = threading.RLock()  # one caller at a time inside a collective (its message scratch is the group's)

    @staticmethod
    def _bind(dev):
        _hip._tls.dev = dev
        dev.sync()  # binds the thread's current HIP device to the member's

    # -- running work on the members ---------------------------------------------------------------
    def map(self, fn, members=None):
        """[fn(i) for i in members] with fn(i) on member i's thread; the first failure is raised after ALL have finished
        (nothing is left running on a member when the caller sees the exception)."""
        idx = range(self.n) if members is None else list(members)
        futs = [self._pool[i].submit(fn, i) for i in idx]
        out, err = [], None
        for f in futs:
            try:
                out.append(f.result())
            except BaseException as e:  # noqa: B902 -- re-raised below
                err = err or e
                out.append(None)
        if err is not None:
            raise err
        return out

    def sync(self):
        for m in self.members:
            m.sync()

    # -- the collectives ---------------------------------------------------------------------------
    def _ptr_array(self, ptrs):
        return (ctypes.c_void_p * self.n)(*[_hip._ptr(p) for p in ptrs])

    def allreduce_device(self, ptrs, count, op="sum"):
        """In place on the members' DEVICE float64 buffers ptrs[i] (count each); asynchronous, stream-ordered on every member."""
        _hip._check(self.lib, self.lib.rr_comm_group_allreduce_dev(self._comms, self.n, self._ptr_array(ptrs), int(count), _OPS[op]))

    def broadcast_device(self, ptrs, nbytes, root=0):
        _hip._check(self.lib, self.lib.rr_comm_group_broadcast_dev(self._comms, self.n, self._ptr_array(ptrs), int(nbytes), int(root)))

    def reduce_stats(self, F, stat_ptrs, nrows, wait=True):
        """Sum the members' statistics IN HBM: stat_ptrs[i] = (dG, db, dyty) of member i (upper triangle of dG valid),
        nrows[i] its row count.  Afterwards every member holds the full symmetric G, b, y^T y of all rows.  Returns the summed
        row count (wait=True: waits for member 0's stream)."""
        cnt = stats_count(F)
        for i, m in enumerate(self.members):
            if self._msg[i] is None or self._msg[i].nbytes < cnt * 8:
                if self._msg[i] is not None:
                    self.sync()  # a peer may still be reading it
                    self._msg[i].free()
                self._msg[i] = m.malloc(cnt * 8)
        rows = (ctypes.c_double * self.n)(*[float(r) for r in nrows])
        tot = ctypes.c_double(0.0)
        _hip._check(self.lib, self.lib.rr_comm_group_reduce_stats_dev(
            self._comms, self.n, int(F), self._ptr_array([p[0] for p in stat_ptrs]), self._ptr_array([p[1] for p in stat_ptrs]),
            self._ptr_array([p[2] for p in stat_ptrs]), rows, self._ptr_array(self._msg), ctypes.byref(tot) if wait else None))
        return int(round(tot.value)) if wait else None

    # -- settings that live on a context ---------------------------------------------------------------
    def set_gram_engine(self, name):
        return [m.set_gram_engine(name) for m in self.members]

    def set_deterministic(self, on=True):
        return [m.set_deterministic(on) for m in self.members]

    def close(self):
        if getattr(self, "pid", None) != os.getpid():
            return
        pool, self._pool = getattr(self, "_pool", None), None
        if pool is None:
            return
        for ex in pool:
            ex.shutdown(wait=True)
        for m in self.members:
            try:
                m.sync()
            except Exception:
                pass
        for b in self._msg:
            if b is not None:
                b.free()
        self._msg = [None] * self.n
        for c in self._comms:
            if c:
                self.lib.rr_comm_destroy(c)
        self._comms = None

    def __repr__(self):
        return "DeviceGroup(devices=%s, transport=%r)" % (list(self.devices), self.transport)


_groups = {}
_groups_lock = threading.Lock()


def get_group(devices, transport=None):
    """The process-local DeviceGroup of these devices (created once: contexts, threads and the communicator are reused by
    every fit / prediction of the process, like ``_hip.get_device``'s context)."""
    key = (os.getpid(), resolve_devices(devices), transport or os.environ.get("RR_COMM_TRANSPORT") or "auto")
    with _groups_lock:
        g = _groups.get(key)
        if g is None:
            for k in [k for k in _groups if k[0] != key[0]]:  # a fork's inheritance: not ours to drive
                del _groups[k]
            g = _groups[key] = DeviceGroup(key[1], key[2])
        return g


@atexit.register
def _close_groups():
    for k, g in list(_groups.items()):
        try:
            g.close()
        except Exception:
            pass
    _groups.clear()


class _PerMember(object):
    """A device buffer per member (the posterior covariance each member formed for itself)."""

    def __init__(self, bufs):
        self.bufs = list(bufs)


def _tree_sum(parts):
    """Element-wise sum of equally structured results (floats, arrays, lists of those) in the order given."""
    first = parts[0]
    if isinstance(first, (list, tuple)):
        return [_tree_sum([p[k] for p in parts]) for k in range(len(first))]
    out = first
    for p in parts[1:]:
        out = out + p
    return out


class ShardedFitState(object):
    """The fit state of a basis (``DeviceFitState`` / ``CatFitState`` interface: what ``StandardLinearModel._elbo_resident``
    calls) with the rows of (X, y) resident on the members of a device group, shard i on member i."""

    MIN_ROWS_PER_MEMBER = 2

    def __init__(self, group, states, bounds):
        self.group, self.states, self.bounds = group, states, bounds
        self.F = states[0].F
        self.dev = states[0].dev
        self.N_total = sum(e - s for s, e in bounds)
        self.best_on_device = False

    @classmethod
    def make(cls, basis, X, y, group):
        """None when the basis has no device-resident fit (the estimator then takes its transform / grad route on the
        default device) or when there are fewer rows than members can share."""
        make = getattr(basis, "device_fit_state", None)
        N = X.shape[0]
        if make is None or N < cls.MIN_ROWS_PER_MEMBER * group.n:
            return None
        bounds = [shard_bounds(N, i, group.n) for i in range(group.n)]
        y = np.asarray(y)

        states = [None] * group.n

        def build(i):
            s, e = bounds[i]
            states[i] = make(X[s:e], y[s:e])

        def drop(i):
            st, states[i] = states[i], None
            if st is not None:
                st.release()
        with group._lock:
            try:
                group.map(build)
            except BaseException:  # a member that failed (out of memory, ...) must not leave the others' shards resident
                group.map(drop)
                raise
            if any(st is None for st in states):
                group.map(drop)
                return None
        return cls(group, states, bounds)

    # -- first pass: statistics ------------------------------------------------------------------------
    def gram_device(self, hypers, reduce=None):
        """The statistics of all rows into every member's resident buffers; returns y^T y."""
        if reduce is not None:
            raise ValueError("a device group and a process group (distributed=True) cannot be combined yet")
        with self.group._lock:  # (the message scratch of the collective belongs to the group)
            self.group.map(lambda i: self.states[i].gram_launch(hypers))
            self.N_total = self.group.reduce_stats(self.F, [st._stat_ptrs() for st in self.states],
                                                   [st.nrows for st in self.states])
        st0 = self.states[0]
        return float(st0.dev.download(st0.acc, (1,), np.float64, offset_bytes=(self.F * self.F + self.F) * 8)[0])

    def gram(self, hypers):
        self.gram_device(hypers)
        return self.stats_host()

    def stats_host(self):
        return self.states[0].stats_host()

    def b_host(self):
        return self.states[0].b_host()

    # -- the posterior, replicated ---------------------------------------------------------------------
    def posterior(self, iL, var):
        res = self.group.map(lambda i: self.states[i].posterior(iL, var))
        if any(r is None for r in res):  # not safely positive definite (on any member): the host SVD route for this step
            return None
        return res[0]

    @property
    def dC(self):
        return _PerMember([st.dC for st in self.states])

    def keep_best(self):
        for st in self.states:
            st.keep_best()
        self.best_on_device = True

    def best_covariance(self):
        return self.states[0].best_covariance()

    # -- second pass -----------------------------------------------------------------------------------
    def second_pass(self, hypers, m, C, var):
        Cs = C.bufs if isinstance(C, _PerMember) else [C] * self.group.n
        res = self.group.map(lambda i: self.states[i].second_pass(hypers, m, Cs[i], var))
        return float(sum(r[0] for r in res)), _tree_sum([r[1] for r in res])

    def release(self):
        self.group.map(lambda i: self.states[i].release())


def sharded_fit_state(basis, X, y, devices):
    return ShardedFitState.make(basis, X, y, get_group(devices))


def gram(basis, X, y, hypers, devices):
    """(Phi^T Phi, Phi^T y, y^T y) of all rows, the row shards on the members of ``devices`` (``basis.gram(..., devices=)``);
    None when the basis has no device-resident route."""
    st = sharded_fit_state(basis, X, np.zeros(X.shape[0]) if y is None else y, devices)
    if st is None:
        return None
    try:
        G, b, yty = st.gram(hypers)
    finally:
        st.release()
    return (G, None, None) if y is None else (G, b, yty)


def map_rows(group, N, fn, n=None):
    """Concatenated results of fn(i, start, stop) over contiguous row shards, member i's on its thread.  fn returns a tuple
    of arrays with one entry per row (or None: not served -- then None is returned)."""
    n = max(1, min(group.n if n is None else n, N))
    bounds = [shard_bounds(N, i, n) for i in range(n)]
    parts = group.map(lambda i: fn(i, *bounds[i]), members=range(n))
    if any(p is None for p in parts):
        return None
    return tuple(None if parts[0][k] is None else np.concatenate([p[k] for p in parts]) for k in range(len(parts[0])))


class _GroupGathered(object):
    """A minibatch made ready on the members' devices ahead of its step (ShardedMinibatchFeatures.prefetch_batch)."""

    def __init__(self, parts, got, key):
        self.parts, self.got, self.key = parts, got, key


class ShardedMinibatchFeatures(object):
    """``basis_functions.MinibatchFeatures`` -- what ``GeneralizedLinearModel._elbo`` programs against -- with the resident
    rows of X sharded over the members of a device group (``GeneralizedLinearModel(devices=[...])``).

    A minibatch is the optimiser's own: row INDICES into all rows (``gen_batch``'s permutation stream, unchanged); member i
    serves the indices that fall into its shard -- its features (gathered from its resident rows), ``fs = Phi ws``, the
    likelihood terms and the three products of the step on ITS rows -- and everything a step returns is a sum over rows
    (``Edm``, ``EdC``, the log-likelihood sums, ``-(EdPhi o dPhi).sum()`` per parameter: glm.py:229-311), added here over the
    members in member order.  The reparameterisation draws do not depend on rows: every member gets the same ones (the host
    array, or the device generator's (seed, step) key).  Minibatches too small to be worth splitting (fewer than
    ``MIN_ROWS_PER_MEMBER`` rows per member: the reference's default batch_size is 10) go to as few members as that allows."""

    MIN_ROWS_PER_MEMBER = 2048
    supports_objective_only = True
    accepts_device_draws = False   # a draw buffer lives on ONE device; members take the host array

    def __init__(self, basis, group, batch_size=None):
        from .basis_functions import MinibatchFeatures
        self.basis, self.group = basis, group
        # members that take part: as many as leave MIN_ROWS_PER_MEMBER rows of a minibatch each (all of them for prediction)
        self.n_use = group.n if batch_size is None else max(1, min(group.n, int(batch_size) // self.MIN_ROWS_PER_MEMBER))
        self.feats = group.map(lambda i: MinibatchFeatures(basis), members=range(self.n_use))
        self.resident = False
        self._parts, self._part_targets = None, {}

    # -- which member serves which rows of a minibatch ---------------------------------------------------
    def make_resident(self, X):
        N = X.shape[0]
        if N < self.n_use * 2:
            return False
        self.bounds = [shard_bounds(N, i, self.n_use) for i in range(self.n_use)]
        self.ends = np.array([e for _, e in self.bounds])
        ok = self.group.map(lambda i: self.feats[i].make_resident(X[self.bounds[i][0]:self.bounds[i][1]]), members=range(self.n_use))
        if not all(ok):
            self.group.map(lambda i: self.feats[i]._drop_children(), members=range(self.n_use))
            return False
        self.resident = True
        return True

    def _split_idx(self, idx):
        """[(member, positions in the minibatch, member-local row indices)] for the members that get rows."""
        idx = np.asarray(idx)
        owner = np.searchsorted(self.ends, idx, side="right")
        parts = []
        for i in range(self.n_use):
            pos = np.nonzero(owner == i)[0]
            if pos.size:
                parts.append((i, pos, idx[pos] - self.bounds[i][0]))
        return parts

    def _split_rows(self, M):
        """Contiguous chunks of a minibatch that is NOT resident (its rows arrive with the step)."""
        n = max(1, min(self.n_use, M // self.MIN_ROWS_PER_MEMBER))
        return [(i, np.arange(*shard_bounds(M, i, n)), None) for i in range(n)]

    def _take(self, arr, pos):
        return None if arr is None else np.ascontiguousarray(np.asarray(arr)[pos])

    # -- the step ------------------------------------------------------------------------------------------
    def stage_targets(self, y, rowarg):
        self._staged = (y, rowarg)

    def take_prefetched_targets(self, gathered, y, rowarg):
        return False

    def prefetch_batch(self, ups, idx, y, rowarg):
        """`MinibatchFeatures.prefetch_batch` per member, for a FUTURE step of the group's resident loop (glm.
        _GroupResidentLoop), on the minibatch worker thread: member i's share of the indices, its rows gathered, its targets --
        through ups[i], an upload context on member i's GPU."""
        parts = self._split_idx(idx)
        got = [None] * self.n_use
        for i, pos, local in parts:
            with _hip.device_scope(self.group.members[i]):
                got[i] = self.feats[i].prefetch_batch(ups[i], local, self._take(y, pos), self._take(rowarg, pos))
        return _GroupGathered(parts, got, (id(y), len(y), rowarg is None))

    def assemble_idx(self, idx, hypers, gathered=None):
        self._parts = self._split_idx(idx)
        y, rowarg = self.__dict__.pop("_staged", (None, None))
        self._part_targets = {}

        def run(k):
            i, pos, local = self._parts[k]
            if y is not None:  # the member's targets go up before its feature kernels (the same arrays meet its step below)
                t = self._part_targets[k] = (id(y), self._take(y, pos), self._take(rowarg, pos))
                self.feats[i].stage_targets(t[1], t[2])
            self.feats[i].assemble_idx(local, hypers)
        self._on_parts(run)

    def assemble(self, X, hypers):
        self._parts, self._part_targets = self._split_rows(X.shape[0]), {}
        self._on_parts(lambda k: self.feats[self._parts[k][0]].assemble(X[self._parts[k][1]], hypers))

    def _on_parts(self, fn):
        """fn(k) for every part k, on its member's thread."""
        members = [p[0] for p in self._parts]
        order = {m: k for k, m in enumerate(members)}
        return self.group.map(lambda i: fn(order[i]), members=members)

    def _step(self, name, y, rowarg, args, kwargs):
        def run(k):
            i, pos, _ = self._parts[k]
            t = self._part_targets.get(k)
            yk, rk = (t[1], t[2]) if (t is not None and t[0] == id(y)) else (self._take(y, pos), self._take(rowarg, pos))
            return getattr(self.feats[i], name)(yk, rk, *args, **kwargs)
        res = self._on_parts(run)
        Edm = None if res[0][0] is None else _tree_sum([r[0] for r in res])
        EdC = None if res[0][1] is None else _tree_sum([r[1] for r in res])
        return Edm, EdC, _tree_sum([r[2] for r in res]), _tree_sum([r[3] for r in res])

    def glm_step_sampled(self, y, rowarg, lik, lik_param, m, C, K, L, seed, step, objective_only=False):
        return self._step("glm_step_sampled", y, rowarg, (lik, lik_param, m, C, K, L, seed, step), {"objective_only": objective_only})

    def glm_step_draws(self, y, rowarg, lik, lik_param, m, C, K, L, E, objective_only=False):
        return self._step("glm_step_draws", y, rowarg, (lik, lik_param, m, C, K, L, E), {"objective_only": objective_only})

    def glm_basis_grads(self, X):
        def run(k):
            i, pos, local = self._parts[k]
            return self.feats[i].glm_basis_grads(X if local is not None else X[pos])
        return _tree_sum(self._on_parts(run))

    # -- prediction: latent function samples Phi(X) W, rows sharded --------------------------------------------
    def project(self, X, hypers, W):
        out = map_rows(self.group, X.shape[0], lambda i, s, e: (self.feats[i].project(X[s:e], hypers, W),), n=self.n_use)
        return out[0]

    def release(self):
        self.group.map(lambda i: self.feats[i].release(), members=range(self.n_use))
        self.resident = False
