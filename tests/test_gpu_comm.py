"""The multi-GPU exchange on real hardware: RCCL bound directly through the C ABI (rr_comm_*), no torch in the
process.  The test box has ONE MI355X, so the two-rank cases run two processes on that GPU; RCCL refuses two ranks on
one device of one host, hence every rank claims its own NCCL_HOSTID and the collective takes RCCL's socket transport --
the product code path (id rendezvous, ncclCommInitRank, pack -> ncclAllReduce -> unpack on the context's stream, the
estimators' use of it) is the one an 8-GPU node runs over xGMI."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, normwise

pytestmark = pytest.mark.gpu


def _run_ranks(code, world, tmp_path, timeout=900, extra_env=None):
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   RR_COMM_RDZV="file:%s" % (tmp_path / "rccl.id"), NCCL_HOSTID="rr-test-rank-%d" % r,
                   NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, (o[-1500:], e[-3000:])
    res = []
    for rc, o, e in outs:
        # RCCL writes its warnings to file descriptor 1 from its own threads: the marker may sit mid-line
        m = re.search(r"RESULT(\{.*\})ENDRESULT", o, flags=re.S)
        assert m, (o[-1500:], e[-3000:])
        res.append(json.loads(m.group(1)))
    return res


def test_stats_pack_unpack_kernels_match_host_packing():
    from revrand_amd import _hip, parallel
    dev = _hip.get_device()
    rs = np.random.RandomState(0)
    for F in (1, 5, 256, 257, 700):
        A = rs.randn(F, F)
        G = np.triu(A)  # what the Gram kernels leave: the upper triangle only
        b, yty = rs.randn(F), np.array([3.25])
        dG, db, dt = dev.upload_vector(G.ravel()), dev.upload_vector(b), dev.upload_vector(yty)
        cnt = parallel.stats_count(F)
        assert dev.lib.rr_stats_msg_count(F) == cnt
        dmsg = dev.zeros(cnt * 8)
        _hip._check(dev.lib, dev.lib.rr_stats_pack_dev(dev.ctx, F, dG.ptr, db.ptr, dt.ptr, 17.0, dmsg.ptr))
        msg = dev.download(dmsg, (cnt,), np.float64)
        Gs = G + np.triu(G, 1).T
        assert np.array_equal(msg, parallel.pack_stats(Gs, b, 3.25, 17))
        dG2, db2, dt2 = dev.zeros(F * F * 8), dev.zeros(F * 8), dev.zeros(8)
        _hip._check(dev.lib, dev.lib.rr_stats_unpack_dev(dev.ctx, F, dmsg.ptr, dG2.ptr, db2.ptr, dt2.ptr))
        assert np.array_equal(dev.download(dG2, (F, F), np.float64), Gs)
        assert np.array_equal(dev.download(db2, (F,), np.float64), b)
        assert dev.download(dt2, (1,), np.float64)[0] == 3.25
        G3, b3, t3, n3 = parallel.unpack_stats(msg, F)
        assert np.array_equal(G3, Gs) and np.array_equal(b3, b) and t3 == 3.25 and n3 == 17


_ONE_RANK = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
os.environ["RR_POSDEF"] = "device"   # F = 120 here: below the default threshold for the device posterior
from revrand_amd import parallel
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel as SLM
comm = parallel.init_rccl_from_env()
parallel.set_comm(comm)
assert (comm.rank, comm.world, comm.kind) == (0, 1, "rccl") and parallel.device_allreduce_available()
ver, path = parallel.RcclComm.load()
rs = np.random.RandomState(0)
X = rs.randn(3000, 5); y = np.sin(X @ rs.randn(5)) + 0.1 * rs.randn(3000)
out = {"rccl": [ver, path]}
for tag, distributed in (("dist", True), ("single", False)):
    basis = bs.RandomRBF(nbases=60, Xdim=5, random_state=1, lenscale=Parameter(np.ones(5), Positive()))
    slm = SLM(basis, distributed=distributed)
    slm.obj_ = -np.inf
    slm._state = basis.device_fit_state(X, y)
    calls = []
    orig = comm.reduce_stats_device
    comm.reduce_stats_device = lambda F, *a, **k: (calls.append(F), orig(F, *a, **k))[1]
    f, (gv, gr, gh) = slm._elbo(X, y, 0.3, 1.2, np.linspace(0.8, 1.3, 5))
    comm.reduce_stats_device = orig
    assert calls == ([120] if distributed else []), calls
    slm._state.release()
    out[tag] = [float(f), float(gv), float(gr)] + np.asarray(gh).tolist() + slm.weights_.tolist()
assert "torch" not in sys.modules
comm.close()
sys.stdout.flush(); print("\nRESULT" + json.dumps(out) + "ENDRESULT", flush=True)
'''


def test_distributed_elbo_over_rccl_keeps_statistics_and_posterior_in_hbm(tmp_path):
    """StandardLinearModel(distributed=True) on an RCCL communicator of one rank: the statistics are packed, all-reduced
    and unpacked IN HBM on the context's stream and the posterior stays on the device; equal to the non-distributed
    evaluation.  No torch in the process."""
    res = _run_ranks(_ONE_RANK % ROOT, 1, tmp_path)[0]
    assert normwise(np.array(res["dist"]), np.array(res["single"])) < 1e-6
    assert res["rccl"][0] >= 20000 and "rccl" in res["rccl"][1]


_TWO_RANKS = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
os.environ["RR_POSDEF"] = "device"
from revrand_amd import _hip, parallel
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel as SLM
comm = parallel.init_rccl_from_env()
parallel.set_comm(comm)
rank, world = comm.rank, comm.world
assert world == 2 and rank == int(os.environ["RANK"])
out = {"rank": rank}
# small host collectives
out["sum"] = comm.allreduce_host(np.array([1.0 + rank, 10.0])).tolist()
out["max"] = comm.allreduce_host(np.array([float(rank), -float(rank)]), op="max").tolist()
out["min"] = comm.allreduce_host(np.array([float(rank), -float(rank)]), op="min").tolist()
box = np.arange(7, dtype=np.int64) * (1 if rank == 0 else 0)
out["bcast"] = comm.broadcast_host(box, root=0).tolist()
comm.barrier()
# the one exchange step: statistics of the two row shards summed in HBM == statistics of all rows
rs = np.random.RandomState(0)
N, d, n = 5001, 6, 160
X = rs.randn(N, d).astype(np.float32); y = (np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)).astype(np.float32)
a, b = parallel.shard_bounds(N, rank, world)
basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
st = basis.device_fit_state(X[a:b], y[a:b])
yty = st.gram_device(np.ones(d), comm.reduce_stats_device)
G, bb, yty2 = st.stats_host()
st.release()
Gf, bf, ytyf = basis.gram(X, y, np.ones(d))
out["stats_err"] = [float(np.abs(G - Gf).max() / np.abs(Gf).max()), float(np.abs(bb - bf).max() / np.abs(bf).max()),
                    abs(yty - ytyf) / ytyf, abs(yty2 - yty)]
out["N_total"] = st.N_total
out["G_sym"] = bool(np.array_equal(G, G.T))
out["G_sum"] = float(G.sum())
# one distributed _elbo and a short distributed fit: every rank walks the same path
slm = SLM(basis, var=Parameter(0.1, Positive()), nstarts=0, maxiter=8, distributed=True, random_state=0)
slm.obj_ = -np.inf
slm._state = slm._make_state(X[a:b], y[a:b])
f, (gv, gr, gh) = slm._elbo(X[a:b], y[a:b], 0.3, 1.2, np.linspace(0.8, 1.3, d))
slm._state.release(); slm._state = None
out["elbo"] = [float(f), float(gv), float(gr)] + np.asarray(gh).tolist()
slm.fit(X[a:b], y[a:b])
out["fit"] = [float(slm.var_), float(slm.regularizer_)] + np.asarray(slm.hypers_).tolist() + [float(slm.obj_)]
if rank == 0:  # the same evaluation in one process on all rows
    s1 = SLM(basis, var=Parameter(0.1, Positive()), nstarts=0, maxiter=8, random_state=0)
    s1.obj_ = -np.inf
    parallel.set_comm(parallel.SingleComm())
    s1._state = s1._make_state(X, y)
    f1, (gv1, gr1, gh1) = s1._elbo(X, y, 0.3, 1.2, np.linspace(0.8, 1.3, d))
    s1._state.release(); s1._state = None
    out["elbo_single"] = [float(f1), float(gv1), float(gr1)] + np.asarray(gh1).tolist()
    parallel.set_comm(comm)
# BASELINE config 3's structure (RandomMatern52 + LinearBasis, rows sharded, RCCL Gram all-reduce) in miniature: one
# distributed _elbo of the concatenation on both ranks, and the all-rows evaluation on rank 0
cat = bs.RandomMatern52(nbases=150, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive())) \
    + bs.LinearBasis(onescol=True)
sc = SLM(cat, distributed=True)
sc.obj_ = -np.inf
sc._state = sc._make_state(X[a:b], y[a:b])
assert type(sc._state).__name__ == "CatFitState"
fc, (gvc, grc, ghc) = sc._elbo(X[a:b], y[a:b], 0.3, [1.2, 0.8], np.linspace(0.8, 1.3, d))
sc._state.release(); sc._state = None
out["cat_elbo"] = [float(fc), float(gvc)] + np.asarray(grc).tolist() + np.asarray(ghc).tolist()
if rank == 0:
    parallel.set_comm(parallel.SingleComm())
    s3 = SLM(cat)
    s3.obj_ = -np.inf
    s3._state = s3._make_state(X, y)
    f3, (gv3, gr3, gh3) = s3._elbo(X, y, 0.3, [1.2, 0.8], np.linspace(0.8, 1.3, d))
    s3._state.release(); s3._state = None
    out["cat_elbo_single"] = [float(f3), float(gv3)] + np.asarray(gr3).tolist() + np.asarray(gh3).tolist()
    parallel.set_comm(comm)
# the GLM's SVI over the same communicator: unequal shards (2501 / 2500 rows), random starts -> identical parameters
from revrand_amd.glm import GeneralizedLinearModel
from revrand_amd.likelihoods import Gaussian, Poisson
yp = np.random.RandomState(5).poisson(np.exp(0.3 * np.sin(X[:, 0].astype(float)))).astype(float)
# -- with the loop resident on every rank (rr_glm_sgd_dist_step: dT and [Edm | EdC | sums | llconst | rows] all-reduced over the ranks
# in HBM, the update replicated) and, from the same seeds, with the host loop around `_elbo` (one host all-reduce per step)
dist_steps = [0]
real_step = _hip.ResidentSgd.step
def spy(self, *a, **k):
    dist_steps[0] += k.get("comm") is not None
    return real_step(self, *a, **k)
_hip.ResidentSgd.step = spy
for tag, resident, lik, yy in (("glm", True, Poisson(), yp), ("glm_host", False, Poisson(), yp),
                               ("glm_gauss", True, Gaussian(), y.astype(float)), ("glm_gauss_host", False, Gaussian(), y.astype(float))):
    glm = GeneralizedLinearModel(lik, bs.RandomRBF(nbases=24, Xdim=d, random_state=3, lenscale=Parameter(np.ones(d), Positive())),
                                 K=2, nsamples=6, batch_size=300, maxiter=6, nstarts=3, random_state=4, distributed=True)
    glm._resident_sgd = resident
    np.random.seed(7)
    glm.fit(X[a:b].astype(float), yy[a:b])
    out[tag] = glm.weights_.ravel().tolist() + glm.covariance_.ravel().tolist() + np.asarray(glm.basis_hypers_).tolist() \
        + [float(glm.regularizer_)] + np.atleast_1d(np.asarray(glm.like_hypers_, dtype=float)).tolist()
out["glm_dist_steps"] = dist_steps[0]
assert "torch" not in sys.modules
comm.barrier()
comm.close()
sys.stdout.flush(); print("\nRESULT" + json.dumps(out) + "ENDRESULT", flush=True)
'''


def test_two_rccl_ranks_sum_shard_statistics_and_fit_identically(tmp_path):
    res = sorted(_run_ranks(_TWO_RANKS % ROOT, 2, tmp_path), key=lambda r: r["rank"])
    assert [r["rank"] for r in res] == [0, 1]
    for r in res:
        assert r["sum"] == [3.0, 20.0] and r["max"] == [1.0, 0.0] and r["min"] == [0.0, -1.0]
        assert r["bcast"] == list(range(7))
        assert r["N_total"] == 5001 and r["G_sym"]
        assert max(r["stats_err"][:3]) < 5e-6 and r["stats_err"][3] == 0.0, r["stats_err"]
    assert res[0]["G_sum"] == res[1]["G_sum"]            # bitwise-identical reduced statistics on both ranks
    assert res[0]["elbo"] == res[1]["elbo"] and res[0]["fit"] == res[1]["fit"]   # ... hence identical paths
    assert normwise(np.array(res[0]["elbo"]), np.array(res[0]["elbo_single"])) < 2e-4   # f32 statistics, two orders
    assert res[0]["cat_elbo"] == res[1]["cat_elbo"]                                      # concatenation (config 3's form)
    assert normwise(np.array(res[0]["cat_elbo"]), np.array(res[0]["cat_elbo_single"])) < 5e-4
    assert res[0]["glm"] == res[1]["glm"] and len(set(np.round(res[0]["glm"], 10))) > 4  # SVI: ranks bit-identical
    # ... through the resident loop (6 steps of rr_glm_sgd_dist_step per fit) as through the host loop, and the two agree
    assert res[0]["glm_dist_steps"] == res[1]["glm_dist_steps"] == 12
    for tag in ("glm", "glm_gauss"):
        assert res[0][tag] == res[1][tag] and res[0][tag + "_host"] == res[1][tag + "_host"]
        assert normwise(np.array(res[0][tag]), np.array(res[0][tag + "_host"])) < 1e-4, tag


def test_bench_two_ranks_as_the_driver_calls_it():
    """`python bench.py --gpus 2` (no launcher): starts its own ranks, exits 0, ONE JSON line, n_gpus as RCCL reports."""
    # (the single-process run inside the N > 1 line is rehearsed at 4 and 8 ranks below and by test_gpu_multigpu.py)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--rows", "600000", "--dist-rows", "20000", "--configs", "elbo"], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, RR_BENCH_NO_SINGLE_PROCESS="1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["exchange"]["ranks_rccl_reports"] == 2
    assert out["exchange"]["message_bytes"] == 8 * (4096 * 4097 // 2 + 4096 + 2)
    assert out["config"]["trace_rel_err"] < 1e-6 and out["value"] > 0
    assert out["configs"]["elbo_rbf_f4096_dist"]["parity"]["ranks_identical"] and "C3_matern52_linear_dist" not in out["configs"]


def test_bench_under_torch_distributed_run_uses_rccl_directly():
    """The driver's documented multi-GPU launch: `python -m torch.distributed.run ... bench.py --gpus N` (N = 1 here, with
    the exchange forced on): the launcher's environment is used, the collective is still the library's own RCCL binding."""
    pytest.importorskip("torch")
    env = dict(os.environ, RR_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "1", "--warmup", "1", "--rows", "500000", "--no-cpu-baseline",
                        "--configs", "none"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["exchange"]["ranks_rccl_reports"] == 1 and out["config"]["trace_rel_err"] < 1e-6


def test_bench_two_ranks_under_torch_distributed_run_as_the_driver_launches_n_gpus():
    """The driver's N > 1 launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` with N = 2 (on this box's one GPU: rank r on device r % 1, socket transport):
    ONE JSON line from rank 0, n_gpus as RCCL reports, the row-sharded `_elbo` after the headline."""
    pytest.importorskip("torch")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "1", "--rows", "600000", "--dist-rows", "20000",
                        "--configs", "elbo"], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, RR_BENCH_NO_SINGLE_PROCESS="1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 == out["exchange"]["ranks_rccl_reports"] and out["config"]["trace_rel_err"] < 1e-6
    assert out["config"]["rows_per_gpu"] == 300000 and out["exchange"]["oversubscribed"]
    c = out["configs"]["elbo_rbf_f4096_dist"]
    assert "error" not in c and c["parity"]["ranks_identical"] and c["parity"]["N_total"] == 20000


def _bench(argv, env=None, timeout=1800):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       timeout=timeout, env=env)
    return r


def _keys(o, prefix=""):
    out = set()
    if isinstance(o, dict):
        for k, v in o.items():
            out.add(prefix + k)
            out |= _keys(v, prefix + k + ".")
    return out


@pytest.mark.parametrize("world", [4, 8])
def test_bench_rehearsal_of_the_drivers_multi_gpu_call(world):
    """`python bench.py --gpus 4` / `--gpus 8` exactly as the driver issues it, oversubscribed on this box's one GPU (rank r ->
    device r % visible GPUs, 8-way file rendezvous, one RCCL all-reduce per step) -- with fewer rows than BASELINE's: after the
    headline the ranks time BASELINE config 3 (RandomMatern52 + LinearBasis, F_tot = 8257, the 273 MB exchange) and the
    RandomRBF F = 4096 `_elbo`, rows sharded (slm.py:142-199 over all shards), every stage on the ranks' clocks."""
    # (world 4: half the rows -- the suite's budget; the 8-way run keeps the larger shapes)
    rows, drows = (800000, 48000) if world == 8 else (400000, 24000)
    # (and without the single-process child and the row-sharded config 4 -- F = 16384: 2 GiB of statistics per rank, all ranks
    # on this box's one GPU -- which the 8-way run covers)
    env = None if world == 8 else dict(os.environ, RR_BENCH_NO_SINGLE_PROCESS="1")
    extra = [] if world == 8 else ["--configs", "c3,elbo"]
    r = _bench(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--rows", str(rows), "--dist-rows", str(drows)] + extra, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 8192, len(lines[0])          # the driver keeps a 9 KB tail of stdout
    out = json.loads(lines[0])
    F = 4096
    assert out["n_gpus"] == world == out["exchange"]["ranks_rccl_reports"]
    assert out["config"]["trace_rel_err"] < 1e-6
    assert out["exchange"]["message_bytes"] == 8 * (F * (F + 1) // 2 + F + 2)
    assert out["config"]["rows_per_gpu"] == rows // world and out["scaling"] == "strong" and out["value"] > 0
    pr = out["per_rank"]
    kmax, kmin = pr["kernel_ms_per_step_max_min_over_ranks"]
    assert kmax >= kmin > 0 and pr["kernel_ms_per_step_sum_over_ranks"] >= kmax
    assert pr["expected_speedup_model"]["speedup_vs_one_gpu"] > 0
    assert out["exchange"]["ms_per_step_pack_allreduce_unpack"] > 0
    rt = out["config"]["runtime"]
    assert rt["hip_runtime"] and rt["rccl"] == out["exchange"]["rccl"]["version"] >= 20000
    # the row-sharded evaluations
    c4rows = max(4096, int(drows * 0.4194304))   # config 4's N = 4 194 304 scaled like --dist-rows (bench.py: --dist-rows-c4)
    for name, Ft in (("C3_matern52_linear_dist", 8257), ("elbo_rbf_f4096_dist", 4096), ("C4elbo_fastfood_f16384_dist", 16384)):
        if Ft == 16384 and world != 8:
            assert name not in out["configs"]
            continue
        c = out["configs"][name]
        assert "error" not in c, c
        nrows = c4rows if Ft == 16384 else drows
        assert c["rows"] == nrows and c["rows_per_gpu"] in (nrows // world, nrows // world + 1) and c["F"] == Ft
        assert c["exchange_bytes"] == 8 * (Ft * (Ft + 1) // 2 + Ft + 2)
        st = c["stage_ms"]
        assert all(st[k] > 0 for k in ("statistics", "exchange", "posterior", "second_pass")) and c["ms"] > 0
        par = c["parity"]
        assert par["N_total"] == nrows and par["G_symmetric"] and par["ranks_identical"]
        assert par["trace_fourier_block"] < 1e-5 and par["neg_elbo_256_rows"] < 1e-5 and par["gradient_256_rows"] < 1e-3
        assert 0 < c["roofline"]["frac"] < 1 and c["speedup_model"]["speedup"] > 0
    # the preflight: where every rank's GPU sits, config 3's message through the communicator before anything is timed
    ex = out["exchange"]
    assert ex["oversubscribed"] and ex["distinct_gpus"] == 1 and len(ex["placement"]) == world
    assert all(p["pci"] == ex["placement"][0]["pci"] and p["rank"] == r for r, p in enumerate(ex["placement"]))
    assert ex["preflight"]["message_bytes"] == 8 * (8257 * 8258 // 2 + 8257 + 2) and ex["preflight"]["busbw_GBps"] > 0
    # the same GPUs behind ONE process (StandardLinearModel(devices=N)'s device group), run by rank 0 as a child once the
    # ranks are done: here the members share the one GPU and take the peer transport
    if world != 8:
        return
    # config 5's SVI step between the ranks: the loop resident on every rank (rr_glm_sgd_dist_step), same bits everywhere
    g5 = out["configs"]["C5_glm_svi_step_dist"]
    assert "error" not in g5 and g5["resident_loop"] and g5["parity"]["ranks_identical"] and g5["ms"] > 0, g5
    sp = out["configs"]["single_process"]
    assert "error" not in sp, sp
    assert sp["n_gpus"] == world and sp["value"] > 0 and sp["members_bit_identical"]
    assert sp["exchange"]["transport"] == "peer" and sp["exchange"]["oversubscribed"]
    assert sp["elbo"]["parity"]["neg_elbo_256_rows"] < 1e-5 and sp["elbo"]["parity"]["gradient_256_rows"] < 1e-3
    assert sp["elbo"]["parity"]["members_bit_identical"]
    assert "error" not in sp["glm_c5"] and sp["glm_c5"]["parity"]["params_after_8_steps_vs_one_context"] < 1e-4, sp["glm_c5"]


def test_bench_launcher_timeout_ends_all_ranks_and_says_which():
    import time
    t0 = time.time()
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--rows", "600000", "--launch-timeout", "0.5"], timeout=300)
    assert r.returncode == 124 and time.time() - t0 < 120, (r.returncode, r.stderr[-2000:])
    assert "still running after --launch-timeout" in r.stderr and r.stdout.strip() == ""


def test_bench_one_gpu_line_has_the_same_keys_under_a_launcher_environment():
    """SCALE's N = 1 point (launcher environment, WORLD_SIZE = 1) must be the BENCH line: same keys, no exchange."""
    argv = ["--gpus", "1", "--steps", "1", "--warmup", "1", "--rows", "500000", "--no-cpu-baseline", "--configs", "none",
            "--no-alt-engine"]
    plain = _bench(argv)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29544")
    launched = _bench(argv, env=env)
    for r in (plain, launched):
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    a, b = (json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0]) for r in (plain, launched))
    assert _keys(a) == _keys(b) and "exchange" not in a and a["n_gpus"] == b["n_gpus"] == 1
    assert a["config"]["runtime"]["hip_runtime"] == b["config"]["runtime"]["hip_runtime"] != ""
    assert a["metric"] == b["metric"] and a["config"]["workload"] == b["config"]["workload"]


def test_bench_side_configuration_that_hangs_does_not_take_the_headline_line():
    """A side configuration stuck in a call that never returns: the watchdog dumps the stacks, writes the ONE JSON line with
    the headline and the configurations finished so far, marks the hung one, and the process exits 0."""
    env = dict(os.environ, RR_BENCH_TEST_HANG="posterior_F8257")
    r = _bench(["--rows", "300000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-alt-engine", "--no-parity-check",
                "--configs", "posterior_f4096,posterior_f8257,predict_moments_n300k", "--config-timeout", "20"], env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["config"]["trace_rel_err"] < 1e-6
    assert out["configs"]["posterior_F4096"]["ms"] > 0
    assert "timed out" in out["configs"]["posterior_F8257"]["error"] and "predict_moments_n300k" not in out["configs"]
    assert "bench.py: config posterior_F8257 timed out" in r.stderr and "Thread" in r.stderr
