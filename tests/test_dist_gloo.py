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
    G = G + G.T  # a Gram matrix: the message carries its upper triangle only
    msg = parallel.pack_stats(G, b, 3.5, 17)
    assert msg.shape == (parallel.stats_count(5),) == (5 * 6 // 2 + 5 + 2,)
    assert np.array_equal(msg[:5], G[0]) and np.array_equal(msg[5:9], G[1, 1:])  # row i = G[i, i:]
    G2, b2, t2, n2 = parallel.unpack_stats(msg, 5)
    assert np.array_equal(G, G2) and np.array_equal(b, b2) and t2 == 3.5 and n2 == 17


def test_id_rendezvous_file_and_tcp(tmp_path):
    """The 128-byte RCCL id goes from rank 0 to the others through a file or a TCP socket (no torch)."""
    import threading
    ident = bytes(range(128))
    for rdzv in ("file:%s" % (tmp_path / "id"), None):
        if rdzv is None:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
            rdzv = "tcp:127.0.0.1:%d" % port
        got = {}

        def run(r):
            got[r] = parallel.exchange_id(r, 3, lambda: ident, rdzv, timeout=60)
        ts = [threading.Thread(target=run, args=(r,)) for r in (2, 1, 0)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(60)
        assert got == {0: ident, 1: ident, 2: ident}


def test_id_rendezvous_ignores_what_a_dead_job_left_behind(tmp_path):
    """a stale id file (its writer is gone) must never be handed to a reader; rank 0 replaces it."""
    import struct
    import subprocess
    import sys
    import threading
    where = tmp_path / "id"
    p = subprocess.Popen([sys.executable, "-c", "pass"])
    p.wait()  # a pid that is certainly dead
    stale = bytes([7]) * 128
    where.write_bytes(parallel._ID_MAGIC + struct.pack("<qq", p.pid, 12345) + stale)
    assert parallel._read_id_file(str(where)) is None
    where.write_bytes(stale)  # the round-2 format (bare 128 bytes): not accepted either
    assert parallel._read_id_file(str(where)) is None
    with pytest.raises(TimeoutError):
        parallel.exchange_id(1, 2, None, "file:%s" % where, timeout=0.3)
    ident, got = bytes(range(128)), {}

    def reader():
        got[1] = parallel.exchange_id(1, 2, None, "file:%s" % where, timeout=30)
    t = threading.Thread(target=reader)
    t.start()
    import time
    time.sleep(0.2)  # the reader polls the stale file meanwhile
    got[0] = parallel.exchange_id(0, 2, lambda: ident, "file:%s" % where, timeout=30)
    t.join(30)
    assert got == {0: ident, 1: ident}
    # a symlink planted under the id's name is not followed by readers
    where.unlink()
    target = tmp_path / "elsewhere"
    target.write_bytes(where.read_bytes() if where.exists() else b"x")
    os.symlink(str(target), str(where))
    assert parallel._read_id_file(str(where)) is None


def test_id_rendezvous_tcp_serves_each_rank_once_and_ignores_strangers():
    import threading
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    rdzv, ident, got = "tcp:127.0.0.1:%d" % port, bytes(range(128)), {}

    def run(r):
        got[r] = parallel.exchange_id(r, 3, lambda: ident, rdzv, timeout=60)
    t0 = threading.Thread(target=run, args=(0,))
    t0.start()
    time.sleep(0.3)
    for junk in (b"", b"GET / HTTP/1.0\r\n\r\n", b"RRCCLREQ" + bytes(8)):  # a port scanner, a wrong world size
        c = socket.create_connection(("127.0.0.1", port), timeout=5)
        c.sendall(junk)
        c.close()
    #  a client whose receive timed out: it asked as rank 1, got no chance to read the reply and hung up without
    # acknowledging -- rank 1 is NOT served by that, its retry below must still be answered
    import struct
    c = socket.create_connection(("127.0.0.1", port), timeout=5)
    c.sendall(struct.pack("<8sii", b"RRCCLREQ", 3, 0)[:12] + struct.pack("<i", 1))
    c.close()
    ts = [threading.Thread(target=run, args=(r,)) for r in (1, 2)]
    for t in ts:
        t.start()
    for t in ts + [t0]:
        t.join(60)
    assert got == {0: ident, 1: ident, 2: ident}


def test_default_rendezvous_is_private_and_goes_tcp_across_nodes(monkeypatch):
    monkeypatch.delenv("RR_COMM_RDZV", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    r = parallel.default_rendezvous()
    assert r.startswith("file:") and os.stat(os.path.dirname(r[5:])).st_mode & 0o077 == 0
    monkeypatch.setenv("WORLD_SIZE", "16")
    monkeypatch.setenv("MASTER_ADDR", "node0")
    monkeypatch.setenv("MASTER_PORT", "29500")
    assert parallel.default_rendezvous() == "tcp:node0:29501"


def test_get_comm_never_imports_torch():
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from revrand_amd import parallel\n"
            "c = parallel.get_comm()\n"
            "assert c.world == 1 and not c.device_reduce and c.allreduce_host([1.0, 2.0]).tolist() == [1.0, 2.0]\n"
            "assert 'torch' not in sys.modules\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


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


@pytest.mark.timeout(300)
def test_three_rank_gloo_allreduce_uneven_shards():
    """An odd world size with N = 1001 rows (shards of 334 / 334 / 333): the single packed exchange still reproduces the
    one-process statistics on every rank."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1, 2]
    assert all(r[1] == 1001 for r in res)
    assert all(r[2] < 1e-12 for r in res)
    assert res[0][3] == res[1][3] == res[2][3]


class _OracleState(object):
    """Test double for revrand_amd.basis_functions.DeviceFitState: the same interface, the per-rank
    device computation replaced by the NumPy oracle (no GPU here).  Everything around it -- the
    model's `_elbo_resident`, packing, the two all-reduces, L-BFGS -- is the product's code."""

    def __init__(self, W, X, y):
        self.W, self.X, self.y = W, X, y

    def gram(self, ls):
        return orc.rff_gram_chunked(self.X, self.y, self.W, ls)

    def second_pass(self, ls, m, C, var):
        Phi = orc.rff_transform(self.X, self.W, ls)
        dP = orc.rff_grad(self.X, self.W, ls)
        err = self.y - Phi @ m
        slabs = [dP] if dP.ndim == 2 else [dP[:, :, i] for i in range(dP.shape[2])]
        dh = [-(m @ (err @ g) - ((g.T @ Phi) * C).sum()) / var for g in slabs]
        return float(err @ err), (dh[0] if dP.ndim == 2 else np.array(dh))

    def release(self):
        pass


def _fit_worker(rank, world, port, q):
    import torch.distributed as dist
    from revrand_amd.basis_functions import RandomRBF
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(3)
        N, d, n = 600, 3, 12
        X = rs.randn(N, d)
        y = np.sin(X @ np.array([1.0, -0.5, 0.3])) + 0.3 * rs.randn(N)
        a, b = parallel.shard_bounds(N, rank, world)
        basis = RandomRBF(nbases=n, Xdim=d, random_state=5, lenscale=Parameter(np.ones(d), Positive()),
                          regularizer=Parameter(1.0, Positive()))
        slm = StandardLinearModel(basis, var=Parameter(0.1, Positive()), nstarts=0, maxiter=10,
                                  distributed=world > 1, random_state=0)
        slm._make_state = lambda Xs, ys: _OracleState(basis.W, Xs, ys)   # no GPU in this container
        # one `_elbo` evaluation at fixed parameters: objective and every gradient
        slm.obj_ = -np.inf
        slm._state = slm._make_state(X[a:b], y[a:b])
        ls = np.array([0.8, 1.1, 1.4])
        f, (g_var, g_reg, g_hyp) = slm._elbo(X[a:b], y[a:b], 0.2, 1.3, ls)
        ev = [float(f), float(g_var), float(g_reg)] + np.asarray(g_hyp).tolist() + slm.weights_.tolist()
        slm._state = None
        # and a short fit: every rank must walk the same path
        slm.fit(X[a:b], y[a:b])
        q.put((rank, ev, float(slm.var_), float(slm.regularizer_), np.asarray(slm.hypers_).tolist(), float(slm.obj_)))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_distributed_fit_equals_single_process():
    """Row-sharded `_elbo` on 2 ranks == the same evaluation in one process (objective, all gradients,
    posterior weights to 1e-9), and a distributed `fit` leaves every rank with identical parameters."""
    import multiprocessing as pymp
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    q1 = pymp.Queue()
    _fit_worker(0, 1, 0, q1)
    single = q1.get(timeout=10)
    assert res[0][1:] == res[1][1:]                       # ranks agree bit-for-bit, evaluation and fit
    assert np.allclose(res[0][1], single[1], rtol=1e-9, atol=1e-9)
    assert res[0][5] > -1e6 and np.isfinite(res[0][2])    # the fit moved to a finite optimum


def _detflag_worker(rank, world, port, q):
    import types
    import torch.distributed as dist
    from revrand_amd.slm import StandardLinearModel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        for flags in ((True, True), (True, False), (False, False)):
            slm = StandardLinearModel(distributed=True)
            slm._state = types.SimpleNamespace(dev=types.SimpleNamespace(deterministic=flags[rank]))
            # asked twice: the agreement is made once (one all-reduce) and cached on the fit state
            out.append([slm._ranks_bit_identical(True), slm._ranks_bit_identical(True), slm._ranks_bit_identical(False),
                        slm._state._deterministic_on_all_ranks])
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ranks_agree_on_the_deterministic_flag_before_dropping_the_result_broadcast():
    """RR_DETERMINISTIC is per process.  The broadcast of rank 0's objective / gradients / posterior is dropped
    only when EVERY rank runs the deterministic device kernels (an all-reduced minimum of the flag, made once per fit
    state) and the step's posterior came from the device -- a rank deciding alone would leave its peers in a collective."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_detflag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == [[True, True, False, True], [False, False, False, False], [False, False, False, False]]


# -- row-sharded SVI of the generalised linear model -------------------------------------------------

class _OracleFeatures(object):
    """Test double for revrand_amd.basis_functions.MinibatchFeatures: the same interface, the per-rank device step
    replaced by the NumPy oracle (no GPU here).  The estimator's `_elbo`, the packing and the all-reduce are the
    product's code."""

    def __init__(self, basis):
        self.basis = basis

    def make_resident(self, X):
        return False

    def assemble(self, X, hypers):
        self.X, self.ls = X, hypers[0]
        self.Phi = orc.rff_transform(X, self.basis.W, self.ls)

    def glm_step_draws(self, y, rowarg, lik, lik_param, m, C, K, L, E):
        D = m.shape[0]
        Edm, EdC, ll = np.empty((D, K)), np.empty((D, K)), np.empty(K)
        self.EdPhi = np.zeros_like(self.Phi)
        for k in range(K):
            e = E[k * L:(k + 1) * L].astype(float)
            Sk = np.sqrt(C[:, k])
            ws = m[:, k] + Sk * e
            fs = ws @ self.Phi.T
            dfs = orc.lik_df("poisson_exp", y, fs)
            Edws = dfs @ self.Phi
            Edm[:, k] = Edws.sum(axis=0) / L
            EdC[:, k] = (Edws * e / Sk).sum(axis=0) / L
            ll[k] = (y * fs - np.exp(fs)).sum()
            self.EdPhi += dfs.T @ ws / (L * K)
        return Edm, EdC, ll, np.zeros(K)

    def glm_basis_grads(self, X):
        dP = orc.rff_grad(self.X, self.basis.W, self.ls)
        return np.array([-(self.EdPhi * dP[:, :, i]).sum() for i in range(dP.shape[2])])

    def release(self):
        pass


def _glm_worker(rank, world, port, q):
    import torch.distributed as dist
    from revrand_amd.basis_functions import RandomRBF
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    from revrand_amd.likelihoods import Poisson
    from revrand_amd.optimize import Adam
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(0)
        N, d, n, K, L = 240, 3, 6, 2, 4
        X = rs.randn(N, d)
        y = rs.poisson(np.exp(0.3 * np.sin(X[:, 0]))).astype(float)
        a, b = parallel.shard_bounds(N, rank, world)
        basis = RandomRBF(nbases=n, Xdim=d, random_state=5, lenscale=Parameter(np.ones(d), Positive()),
                          regularizer=Parameter(1.5, Positive()))
        glm = GeneralizedLinearModel(Poisson(), basis, K=K, nsamples=L, batch_size=b - a, maxiter=6, nstarts=0,
                                     random_state=3, updater=Adam(alpha=0.05), distributed=world > 1)
        glm._mbf = _OracleFeatures(basis)                      # no GPU in this container
        # one evaluation on the whole shard
        glm.B_, glm.D_ = 1.0, 2 * n
        glm._GeneralizedLinearModel__it = -1
        m, C = 0.1 * rs.randn(2 * n, K), rs.gamma(2., 0.5, (2 * n, K))
        f, (ndm, ndC, dL, dlp, dbp) = glm._elbo(m, C, 1.5, [], np.array([0.9, 1.1, 1.3]), X[a:b], y[a:b])
        ev = [float(f), float(dL)] + ndm.ravel().tolist() + ndC.ravel().tolist() + np.asarray(dbp).tolist()
        # and a short fit on the shard (the double is re-installed: fit() releases its features at the end)
        glm.random_ = np.random.RandomState(3)
        orig = glm._features
        glm._features = lambda: glm.__dict__.setdefault("_mbf", _OracleFeatures(basis))
        glm.fit(X[a:b], y[a:b])
        q.put((rank, ev, glm.weights_.ravel().tolist(), np.asarray(glm.basis_hypers_).tolist(), float(glm.regularizer_)))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_distributed_glm_step_equals_single_process():
    """Row-sharded SVI: with every rank's minibatch covering its shard, the all-reduced `_elbo` on 2 ranks equals the
    single-process evaluation on all rows (same seed -> same draws), and after a short distributed `fit` the ranks
    hold identical parameters."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_glm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in procs])
    for p in procs:
        p.join(60)
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_glm_worker, args=(0, 1, 0, q1))
    p1.start()
    single = q1.get(timeout=500)
    p1.join(60)
    assert np.allclose(res[0][1], res[1][1], rtol=0, atol=0)              # ranks agree exactly
    assert np.allclose(res[0][1], single[1], rtol=1e-9, atol=1e-10)        # and equal the all-rows evaluation
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] and res[0][4] == res[1][4]


def _glm_unequal_worker(rank, world, port, q):
    """Shards that differ by one row AND random starts: the ranks' RandomStates drift apart (a permutation of N_local
    rows consumes N_local-dependent state), so candidates and start point must be rank 0's (optimize.py)."""
    import torch.distributed as dist
    from revrand_amd.basis_functions import RandomRBF
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    from revrand_amd.likelihoods import Poisson
    from revrand_amd.optimize import Adam
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        np.random.seed(1000 + rank)   # the ranks' GLOBAL streams differ, as in separately started processes
        rs = np.random.RandomState(0)
        N, d, n, K, L = 241, 3, 6, 2, 4
        X = rs.randn(N, d)
        y = rs.poisson(np.exp(0.3 * np.sin(X[:, 0]))).astype(float)
        a, b = parallel.shard_bounds(N, rank, world)
        assert (b - a) == (121 if rank == 0 else 120)
        basis = RandomRBF(nbases=n, Xdim=d, random_state=5, lenscale=Parameter(np.ones(d), Positive()),
                          regularizer=Parameter(1.5, Positive()))
        glm = GeneralizedLinearModel(Poisson(), basis, K=K, nsamples=L, batch_size=50, maxiter=5, nstarts=4,
                                     random_state=3, updater=Adam(alpha=0.05), distributed=True)
        glm._features = lambda: glm.__dict__.setdefault("_mbf", _OracleFeatures(basis))
        glm.fit(X[a:b], y[a:b])
        q.put((rank, glm.weights_.ravel().tolist(), glm.covariance_.ravel().tolist(),
               np.asarray(glm.basis_hypers_).tolist(), float(glm.regularizer_)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_glm_unequal_shards_with_random_starts_stay_identical():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_glm_unequal_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1:] == res[1][1:]          # identical parameters on both ranks, bit for bit
    assert len(set(np.round(res[0][1], 12))) > 4   # and the mixture did not start (or stay) degenerate


def test_draws_made_one_step_ahead_consume_the_stream_like_a_sequential_run():
    """GLM, reference random stream: the worker thread that builds minibatch t+1 also makes its standard-normal draws
    (revrand_amd/glm.py `_draw_ahead`).  The fitted parameters must equal, bit for bit, those of a strictly sequential run in
    which `_elbo` draws for itself (glm.py:300's order: batch_t, e_t, batch_t+1, ...), and random_ must end in the same state."""
    from revrand_amd.basis_functions import RandomRBF
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    from revrand_amd.likelihoods import Poisson
    from revrand_amd.optimize import Adam
    rs = np.random.RandomState(0)
    N, d, n, K, L = 230, 3, 6, 2, 4
    X = rs.randn(N, d)
    y = rs.poisson(np.exp(0.3 * np.sin(X[:, 0]))).astype(float)
    out = []
    for ahead in (True, False):
        np.random.seed(7)  # the start point comes from the global stream
        basis = RandomRBF(nbases=n, Xdim=d, random_state=5, lenscale=Parameter(np.ones(d), Positive()),
                          regularizer=Parameter(1.5, Positive()))
        glm = GeneralizedLinearModel(Poisson(), basis, K=K, nsamples=L, batch_size=60, maxiter=9, nstarts=3,
                                     random_state=3, updater=Adam(alpha=0.05))
        glm._prefetch_draws = ahead
        glm._features = lambda g=glm, b=basis: g.__dict__.setdefault("_mbf", _OracleFeatures(b))
        glm.fit(X, y)
        out.append((glm.weights_.copy(), glm.covariance_.copy(), np.asarray(glm.basis_hypers_).copy(),
                    glm.random_.randint(0, 2 ** 31 - 1)))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
