"""
Row-sharded Gram assembly across processes (one process per GPU) -- SURVEY 8e.

The path shards embarrassingly over rows: every rank accumulates the sufficient statistics of ITS rows -- what
``revrand/slm.py:145-157`` computes from Phi (``Phi.T.dot(Phi)``, ``Phi.T.dot(y)``) plus ``y^T y`` and N -- then ONE
all-reduce sums the message ``[upper triangle of G | b | y^T y | N]`` (F (F + 1) / 2 + F + 2 float64: 67 MB at
F = 4096), after which every rank holds the global statistics and runs the same Cholesky.

Transports (a ``Comm`` object; ``get_comm()`` picks one):

* ``RcclComm``  -- the product transport: RCCL over xGMI, bound directly through the C ABI (``rr_comm_*`` in
  ``include/revrand_hip.h``: ncclGetUniqueId / ncclCommInitRank / ncclAllReduce on the context's stream).  No PyTorch.
  The 128-byte id goes from rank 0 to the others through a file or a TCP socket (``RR_COMM_RDZV``); under
  ``torch.distributed.run`` / ``bench.py``'s own launcher the environment (RANK, WORLD_SIZE, LOCAL_RANK) is enough.
* ``TorchComm`` -- the CPU test transport: a ``torch.distributed`` group the CALLER initialised (``gloo`` in
  ``tests/test_dist_gloo.py``).  torch is only ever touched when such a group already exists in the process.
* ``SingleComm`` -- one rank, every collective a no-op.
"""
import ctypes
import logging
import os
import socket
import sys
import tempfile
import time

import numpy as np

log = logging.getLogger(__name__)

_OPS = {"sum": 0, "max": 1, "min": 2}


def shard_bounds(N, rank, world):
    """Contiguous row block [start, stop) of `rank`; blocks differ by at most one row."""
    base, extra = divmod(int(N), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def stats_count(F):
    """Length of the exchange message for an (F, F) Gram: upper triangle + b + y^T y + N."""
    return F * (F + 1) // 2 + F + 2


def pack_stats(G, b, yty, N):
    """[G[i, i:] for every row i | b | yty | N] as one float64 vector (the message of the one exchange step)."""
    F = G.shape[0]
    out = np.empty(stats_count(F))
    out[:F * (F + 1) // 2] = G[np.triu_indices(F)]
    out[-F - 2:-2] = b
    out[-2] = yty
    out[-1] = N
    return out


def unpack_stats(buf, F):
    """(G full symmetric, b, yty, N) from a packed message."""
    G = np.zeros((F, F))
    iu = np.triu_indices(F)
    G[iu] = buf[:F * (F + 1) // 2]
    G = G + np.triu(G, 1).T
    return G, np.array(buf[-F - 2:-2]), float(buf[-2]), int(round(float(buf[-1])))


# ------------------------------------------------------------------------------------------------
# transports
# ------------------------------------------------------------------------------------------------

class Comm(object):
    """What the estimators and bench.py need from a process group."""
    rank, world = 0, 1
    device_reduce = False  # can sum a buffer in HBM in place (RCCL)
    kind = "single"

    def allreduce_host(self, buf, op="sum"):
        return np.ascontiguousarray(buf, dtype=np.float64)

    def broadcast_host(self, arr, root=0):
        return arr

    def barrier(self):
        pass

    def close(self):
        pass


class SingleComm(Comm):
    pass


class RcclComm(Comm):
    """RCCL communicator of this process's GPU (rr_comm).  `ident`: the 128 bytes of ``unique_id()`` made by ONE rank
    and handed to all (``exchange_id``)."""
    device_reduce = True
    kind = "rccl"

    def __init__(self, rank, world, ident, device=None):
        from . import _hip
        self._hip = _hip
        self.dev = _hip.get_device(device)
        self.lib = self.dev.lib
        if len(ident) != 128:
            raise ValueError("an RCCL unique id has 128 bytes")
        ver, path = RcclComm.load()  # the librccl paired with this process's HIP runtime
        h = ctypes.c_void_p()
        idbuf = ctypes.create_string_buffer(bytes(ident), 128)
        _hip._check(self.lib, self.lib.rr_comm_init_rank(self.dev.ctx, int(rank), int(world), idbuf, ctypes.byref(h)))
        self.h = h
        r, w = ctypes.c_int(), ctypes.c_int()
        _hip._check(self.lib, self.lib.rr_comm_info(h, ctypes.byref(r), ctypes.byref(w)))
        self.rank, self.world = r.value, w.value  # as RCCL reports them
        self._msg = None
        self.rccl_version, self.rccl_library = ver, path
        self._assert_same_rccl()

    def _assert_same_rccl(self):
        """Every rank must have bound the same RCCL VERSION: a mixed job -- one rank on the torch wheel's RCCL, another on
        /opt/rocm's -- may run, and then fail in a collective much later.  One tiny all-reduce.  Only a version mismatch is
        fatal: the same build installed under different prefixes on different nodes (per-node virtualenvs, /opt/rocm-X.Y
        symlink targets) is fine, so differing library FILE NAMES are a logged warning, and an unknown path (the loader
        resolved a bare soname) is not compared at all."""
        if self.world < 2:
            return
        import zlib
        name = os.path.basename(self.rccl_library) if self.rccl_library else ""
        tag = float(zlib.crc32(name.encode())) if name else 0.0
        v = float(self.rccl_version)
        hi = self.allreduce_host(np.array([v, -v, tag, -tag]), op="max")
        if hi[0] != -hi[1]:
            raise RuntimeError("rank %d bound RCCL %d from %s, but the ranks of this job do not all use the same RCCL version "
                               "(%d..%d): start every rank with the same RR_HIP_RUNTIME / RR_RCCL_LIB"
                               % (self.rank, self.rccl_version, self.rccl_library or "(unknown path)", int(-hi[1]), int(hi[0])))
        if hi[2] != -hi[3] and self.rank == 0:
            log.warning("the ranks bound RCCL %d from library files of different names (rank 0: %s): the same version, "
                        "so the job goes on", self.rccl_version, name or "(unknown)")

    @staticmethod
    def load(path=None):
        """Bind librccl (idempotent); returns (version, path)."""
        from . import _hip
        lib = _hip.load_library()
        p = path or _hip.rccl_library_path()
        _hip._check(lib, lib.rr_comm_load(p.encode() if p else None))
        v, buf = ctypes.c_int(), ctypes.create_string_buffer(512)
        _hip._check(lib, lib.rr_comm_version(ctypes.byref(v), buf, 512))
        return v.value, buf.value.decode()

    @staticmethod
    def unique_id():
        from . import _hip
        RcclComm.load()
        lib = _hip.load_library()
        buf = ctypes.create_string_buffer(128)
        _hip._check(lib, lib.rr_comm_unique_id(buf))
        return buf.raw

    def allreduce_host(self, buf, op="sum"):
        buf = np.array(buf, dtype=np.float64, order="C", copy=True).ravel()
        self._hip._check(self.lib, self.lib.rr_comm_allreduce_host(self.h, buf.ctypes.data_as(ctypes.c_void_p), buf.size,
                                                                   _OPS[op]))
        return buf

    def allreduce_device(self, ptr, count, op="sum"):
        """In place on a DEVICE float64 buffer, asynchronous on the context's stream."""
        self._hip._check(self.lib, self.lib.rr_comm_allreduce_dev(self.h, self._hip._ptr(ptr), int(count), _OPS[op]))

    def broadcast_host(self, arr, root=0):
        arr = np.ascontiguousarray(arr)
        self._hip._check(self.lib, self.lib.rr_comm_broadcast_host(self.h, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes,
                                                                   int(root)))
        return arr

    def barrier(self):
        self._hip._check(self.lib, self.lib.rr_comm_barrier(self.h))

    def reduce_stats_device(self, F, pG, pb, pyty, nrows, wait=True):
        """Sum [G | b | yty | N] over the ranks IN HBM: pack the upper triangle, one ncclAllReduce, unpack into the full
        symmetric G -- all on the context's stream (no host synchronisation before it).  Returns the summed N (wait=True)."""
        cnt = stats_count(F)
        if self._msg is None or self._msg.nbytes < cnt * 8:
            if self._msg is not None:
                self.dev.sync()
                self._msg.free()
            self._msg = self.dev.malloc(cnt * 8)
        tot = ctypes.c_double(float(nrows))
        self._hip._check(self.lib, self.lib.rr_comm_reduce_stats_dev(
            self.h, int(F), self._hip._ptr(pG), self._hip._ptr(pb), self._hip._ptr(pyty), float(nrows), self._msg.ptr,
            ctypes.byref(tot) if wait else None))
        return int(round(tot.value)) if wait else None

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            if self._msg is not None:
                self._msg.free()
                self._msg = None
            self.lib.rr_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TorchComm(Comm):
    """A torch.distributed group the caller initialised -- the CPU test transport (gloo).  Host vectors only."""
    kind = "torch"

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.kind = "torch-" + dist.get_backend(group)

    def allreduce_host(self, buf, op="sum"):
        buf = np.array(buf, dtype=np.float64, order="C", copy=True).ravel()
        if self.world > 1:
            rop = {"sum": self._dist.ReduceOp.SUM, "max": self._dist.ReduceOp.MAX, "min": self._dist.ReduceOp.MIN}[op]
            self._dist.all_reduce(self._torch.from_numpy(buf), op=rop, group=self.group)
        return buf

    def broadcast_host(self, arr, root=0):
        arr = np.ascontiguousarray(arr)
        if self.world > 1:
            flat = arr.reshape(-1).view(np.uint8)
            self._dist.broadcast(self._torch.from_numpy(flat), src=root, group=self.group)
        return arr

    def barrier(self):
        if self.world > 1:
            self._dist.barrier(group=self.group)


# ------------------------------------------------------------------------------------------------
# rendezvous of the RCCL id (rank 0 -> everyone): a file or a TCP socket
# ------------------------------------------------------------------------------------------------

_ID_MAGIC = b"RRCCLID1"
_ID_ACK = b"\x06"


def _rdzv_dir():
    """A directory only this user can write (0700, owned by us, not a symlink) for the id files."""
    d = os.path.join(tempfile.gettempdir(), "rr_comm_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("rendezvous directory %s is not a private directory of uid %d" % (d, os.getuid()))
    return d


def _proc_start(pid):
    """Start time (clock ticks since boot) of a live process, or None."""
    try:
        with open("/proc/%d/stat" % pid) as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return None


def default_rendezvous():
    """``RR_COMM_RDZV`` ("file:/path" or "tcp:host:port"), else
    * ranks on more than one node (WORLD_SIZE > LOCAL_WORLD_SIZE): ``tcp:MASTER_ADDR:MASTER_PORT+1`` -- a file is node-local;
    * one node: a file in a private per-user directory whose name every worker of ONE launcher derives identically (the
      launcher's pid -- torch.distributed.run's agent, or bench.py's own launcher -- MASTER_PORT and the restart count).
      The file carries the writer's pid and start time, and readers only accept the id of a LIVE writer (``exchange_id``),
      so what an earlier, crashed job left behind is never used."""
    r = os.environ.get("RR_COMM_RDZV")
    if r:
        return r
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > local_world:
        addr, port = os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")
        if not addr or not port:
            raise RuntimeError("ranks on several nodes need RR_COMM_RDZV=tcp:<host>:<port> (or MASTER_ADDR / MASTER_PORT)")
        return "tcp:%s:%d" % (addr, int(port) + 1)
    key = "%d_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return "file:" + os.path.join(_rdzv_dir(), "rr_comm_%s.id" % key)


def _read_id_file(where):
    """The 128-byte id in `where` if a live process wrote it; None otherwise (absent, partial, stale)."""
    import struct
    try:
        fd = os.open(where, os.O_RDONLY | os.O_NOFOLLOW)
    except OSError:
        return None
    with os.fdopen(fd, "rb") as f:
        blob = f.read()
    if len(blob) != 8 + 16 + 128 or blob[:8] != _ID_MAGIC:
        return None
    pid, start = struct.unpack("<qq", blob[8:24])
    if _proc_start(pid) != start:  # the writer is gone (or its pid was recycled): an earlier job's file
        return None
    return blob[24:]


def exchange_id(rank, world, make_id, rendezvous=None, timeout=600.0):
    """Rank 0 calls make_id() and publishes the bytes; every other rank receives them."""
    import struct
    if world == 1:
        return make_id()
    rdzv = rendezvous or default_rendezvous()
    kind, _, where = rdzv.partition(":")
    deadline = time.time() + timeout
    if kind == "file":
        if rank == 0:
            try:  # whatever an earlier job left under this name
                os.unlink(where)
            except OSError:
                pass
            ident = make_id()
            tmp = where + ".tmp%d" % os.getpid()
            try:
                os.unlink(tmp)
            except OSError:
                pass
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
            with os.fdopen(fd, "wb") as f:
                f.write(_ID_MAGIC + struct.pack("<qq", os.getpid(), _proc_start(os.getpid()) or 0) + ident)
            os.replace(tmp, where)  # atomic: readers see the whole record or nothing
            return ident
        while time.time() < deadline:
            ident = _read_id_file(where)
            if ident is not None:
                return ident
            time.sleep(0.01)
        raise TimeoutError("no RCCL id of a live rank 0 at %s after %.0f s" % (where, timeout))
    if kind == "tcp":
        host, _, port = where.rpartition(":")
        token = struct.pack("<8sii", b"RRCCLREQ", int(world), 0)[:12]
        if rank == 0:
            ident = make_id()
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            # loopback names stay on loopback; any other name is rank 0's address as the OTHER nodes see it: all interfaces
            srv.bind((host if host.startswith("127.") or host == "localhost" else "", int(port)))
            srv.listen(max(world, 8))
            served = set()
            # A rank counts as served when it ACKNOWLEDGED the id (one byte back), not when sendall() returned: a client
            # whose 5 s receive timeout fired while this single-threaded server was held up by junk connections has closed
            # its socket, the send into it may still "succeed", and its retry must then be answered -- so requests of
            # already-served ranks are answered as well, and only distinct acknowledged ranks end the loop.
            while len(served) < world - 1:
                left = deadline - time.time()
                if left <= 0:
                    srv.close()
                    raise TimeoutError("RCCL id server: only ranks %s of %d asked within %.0f s" % (sorted(served), world, timeout))
                srv.settimeout(left)
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                try:
                    c.settimeout(5.0)
                    req = b""
                    while len(req) < 16:
                        part = c.recv(16 - len(req))
                        if not part:
                            break
                        req += part
                    if len(req) == 16 and req[:12] == token:
                        r = struct.unpack("<i", req[12:])[0]
                        if 1 <= r < world:
                            c.sendall(ident)
                            if c.recv(1) == _ID_ACK:
                                # the client holds the id and said so; tell it that its acknowledgement ARRIVED -- a client
                                # that does not hear this (it was descheduled past our receive timeout and we gave up on
                                # it) asks again instead of walking off while we wait for it
                                c.sendall(_ID_ACK)
                                served.add(r)
                except OSError:
                    pass
                finally:
                    c.close()
            srv.close()
            return ident
        while time.time() < deadline:
            try:
                with socket.create_connection((host, int(port)), timeout=5.0) as c:  # closed on every path
                    c.sendall(token + struct.pack("<i", int(rank)))
                    ident = b""
                    while len(ident) < 128:
                        part = c.recv(128 - len(ident))
                        if not part:
                            break
                        ident += part
                    if len(ident) == 128:
                        c.sendall(_ID_ACK)
                        if c.recv(1) == _ID_ACK:  # the server counted us: done.  Otherwise ask again (it answers repeats)
                            return ident
            except OSError:
                pass
            time.sleep(0.05)
        raise TimeoutError("no RCCL id from %s after %.0f s" % (where, timeout))
    raise ValueError("RR_COMM_RDZV must be file:<path> or tcp:<host>:<port>, got %r" % rdzv)


def init_rccl_from_env(device=None, rendezvous=None):
    """The RCCL communicator of a launcher-started process: RANK / WORLD_SIZE / LOCAL_RANK from the environment
    (torch.distributed.run sets them; so does bench.py's own launcher), the id through ``default_rendezvous()``."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    rdzv = rendezvous or default_rendezvous()
    ident = exchange_id(rank, world, RcclComm.unique_id, rdzv)
    comm = RcclComm(rank, world, ident, device)
    if rank == 0 and world > 1 and rdzv.startswith("file:"):
        try:  # ncclCommInitRank returned: every rank has read it
            os.unlink(rdzv[5:])
        except OSError:
            pass
    return comm


_comm = None


def set_comm(comm):
    """Install the process group the estimators use (None: back to auto-detection).  Returns the previous one."""
    global _comm
    prev, _comm = _comm, comm
    return prev


def _torch_group_initialised():
    dist = sys.modules.get("torch.distributed")  # never imports torch
    try:
        return bool(dist is not None and dist.is_available() and dist.is_initialized())
    except Exception:
        return False


def get_comm():
    """The process group of this process: the one given to ``set_comm``; else a torch.distributed group the caller
    initialised (gloo: the CPU test transport; nccl: only its store is used, to hand the RCCL id around); else RCCL
    from the launcher's environment when WORLD_SIZE > 1; else a single rank."""
    global _comm
    if _comm is not None:
        return _comm
    if _torch_group_initialised():
        dist = sys.modules["torch.distributed"]
        if dist.get_backend() == "nccl":
            box = [RcclComm.unique_id() if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            _comm = RcclComm(dist.get_rank(), dist.get_world_size(), box[0])
            return _comm
        return TorchComm()  # not cached: the group may be destroyed and re-created (tests)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        _comm = init_rccl_from_env()
        return _comm
    return SingleComm()


# ------------------------------------------------------------------------------------------------
# what the estimators call
# ------------------------------------------------------------------------------------------------

def allreduce_host(buf, op="sum"):
    """Sum (max, min) a host float64 vector over the ranks and return it."""
    return get_comm().allreduce_host(buf, op)


def allreduce_packed(buf):
    """Sum a packed float64 NumPy message over the ranks, in place."""
    buf[...] = get_comm().allreduce_host(buf).reshape(buf.shape)
    return buf


def device_allreduce_available():
    """True when the ranks' statistics can be summed in HBM (an RCCL communicator)."""
    return bool(get_comm().device_reduce)


def sharded_gram(local_gram, X, y, rank, world):
    """Global (G, b, yty, N) from row shards.

    local_gram(X_rows, y_rows) -> (G, b, yty) is the per-rank device computation
    (e.g. ``lambda Xs, ys: basis.gram(Xs, ys, lenscale)``).  X, y are the FULL arrays (every rank
    slices its own block).
    """
    start, stop = shard_bounds(X.shape[0], rank, world)
    G, b, yty = local_gram(X[start:stop], y[start:stop])
    buf = get_comm().allreduce_host(pack_stats(G, b, yty, stop - start))
    return unpack_stats(buf, G.shape[0])
