"""
Row-sharded Gram assembly across processes (one process per GPU).

The path shards embarrassingly over rows: every rank accumulates ``[G | b | y^T y | N]`` for its
rows, then ONE all-reduce (torch.distributed: ``nccl`` = RCCL over xGMI on the GPUs, ``gloo`` in
the CPU tests) sums the packed buffer, after which every rank holds the global statistics and
runs the same host Cholesky.  torch is plumbing here (process group + collective); the arithmetic
is the HIP library's.
"""
import numpy as np


def shard_bounds(N, rank, world):
    """Contiguous row block [start, stop) of `rank`; blocks differ by at most one row."""
    base, extra = divmod(int(N), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def pack_stats(G, b, yty, N):
    """[G.ravel() | b | yty | N] as one float64 vector (one message)."""
    F = G.shape[0]
    out = np.empty(F * F + F + 2)
    out[:F * F] = G.ravel()
    out[F * F:F * F + F] = b
    out[-2] = yty
    out[-1] = N
    return out


def unpack_stats(buf, F):
    G = np.array(buf[:F * F]).reshape(F, F)
    return G, np.array(buf[F * F:F * F + F]), float(buf[-2]), int(round(float(buf[-1])))


def allreduce_packed(buf, group=None):
    """Sum a packed float64 buffer over the process group, in place.

    `buf` is a torch tensor (CUDA for nccl, CPU for gloo) or a NumPy array (wrapped without a
    copy for gloo).  Returns the same object.  No-op when torch.distributed is not initialised.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return buf
    t = torch.from_numpy(buf) if isinstance(buf, np.ndarray) else buf
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return buf


def allreduce_host(buf, group=None):
    """Sum a host float64 vector over the ranks and return it: directly with gloo; with nccl (= RCCL)
    through a CUDA tensor on this process's GPU, since that backend reduces device memory only."""
    import torch
    import torch.distributed as dist
    buf = np.ascontiguousarray(buf, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return buf
    if dist.get_backend(group) == "nccl":
        t = torch.from_numpy(buf).cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()
    dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM, group=group)
    return buf


def sharded_gram(local_gram, X, y, rank, world, group=None):
    """Global (G, b, yty, N) from row shards.

    local_gram(X_rows, y_rows) -> (G, b, yty) is the per-rank device computation
    (e.g. ``lambda Xs, ys: basis.gram(Xs, ys, lenscale)``).  X, y are the FULL arrays (every rank
    slices its own block) -- pass already-sliced data with rank=0, world=1 semantics by calling
    ``allreduce_packed`` directly when the shards live in different processes' memory.
    """
    start, stop = shard_bounds(X.shape[0], rank, world)
    G, b, yty = local_gram(X[start:stop], y[start:stop])
    buf = allreduce_packed(pack_stats(G, b, yty, stop - start), group)
    return unpack_stats(buf, G.shape[0])


class _DeviceSpan(object):
    """A span of float64 device memory owned by librevrand_hip, exposed through ``__cuda_array_interface__`` so
    that torch can wrap it WITHOUT a copy (torch and the library share one HIP runtime, see _hip.load_library)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_allreduce_available(group=None):
    """True when a device buffer can be summed over the ranks in place: an initialised nccl (= RCCL) group."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:
        return False
    return bool(dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
                and torch.cuda.is_available())


def allreduce_device(ptr, count, group=None):
    """Sum `count` float64 values at device address `ptr` over the ranks, in place, with RCCL.  The caller has
    synchronised its own stream; on return the collective has completed (torch's stream is synchronised)."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(_DeviceSpan(ptr, count), device="cuda")
    if dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    torch.cuda.synchronize()
