"""
The N>1 path on CPU: two processes, gloo backend, row shards -> packed all-reduce -> identical
global statistics on every rank.  The per-rank device computation is
stood in for by the NumPy oracle here (there is no GPU in this container); the sharding, packing
and collective code under test is the product's (revrand_amd/parallel.py), the same functions
bench.py uses with the nccl backend.
"""
import os
import socket

import numpy as np
import pytest

import revrand_oracle as orc
from revrand_amd import parallel


def test_shard_bounds_cover_rows_exactly():
    for N in (0, 1, 7, 10, 1000003):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(N, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip():
    rs = np.random.RandomState(0)
    G, b = rs.randn(5, 5), rs.randn(5)
    G2, b2, t2, n2 = parallel.unpack_stats(parallel.pack_stats(G, b, 3.5, 17), 5)
    assert np.array_equal(G, G2) and np.array_equal(b, b2) and t2 == 3.5 and n2 == 17


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(5)
        N, d, n = 1001, 6, 24
        X = rs.randn(N, d)
        y = np.sin(X @ rs.randn(d))
        W = orc.weights_rbf(d, n, 3)

        def local(Xs, ys):  # stand-in for basis.gram on this rank's GPU
            return orc.rff_gram_chunked(Xs, ys, W, 1.3, chunk=200)

        G, b, yty, Ntot = parallel.sharded_gram(local, X, y, rank, world)
        Gr, br, tr = orc.rff_gram_chunked(X, y, W, 1.3)
        err = max(np.abs(G - Gr).max() / np.abs(Gr).max(), np.abs(b - br).max() / np.abs(br).max(),
                  abs(yty - tr) / tr)
        # identical posterior on every rank after the single exchange
        m, _, _ = orc.slm_posterior_from_stats(G, b, 0.4, np.full(2 * n, 1.0))
        q.put((rank, Ntot, err, float(m.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_allreduce():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] == 1001 for r in res)
    assert all(r[2] < 1e-12 for r in res)
    assert res[0][3] == res[1][3]  # bitwise-identical reduced statistics -> identical weights
