#!/usr/bin/env python3
"""
bench.py -- feature-rows/sec of the hot path (Phi + Phi^T Phi + Phi^T y) on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): RandomRBF, N = 10M rows, D = 32, F = 2*nbases = 4096, f32 arithmetic, synthetic
Gaussian inputs resident in HBM before the timed region.  One "step" is one full pass of the path over this rank's rows:
zero the (F,F)/(F,) accumulators, project + cos/sin + accumulate every row, (N>1: pack the upper triangle, ONE RCCL
all-reduce of [tri G | b | yty | N], unpack) mirror the triangle.

N > 1: one process per GPU, rows sharded across ranks (fixed global N -> "strong").  `python bench.py --gpus N` starts its
own N ranks (RANK / LOCAL_RANK / WORLD_SIZE in the children's environment); under `python -m torch.distributed.run ...
bench.py --gpus N` the launcher's environment is used as is.  Either way the exchange is RCCL bound directly through
librevrand_hip.so's C ABI (rr_comm_*, revrand_amd/parallel.py) -- there is no torch in this process.

Rank 0 prints ONE JSON line with the contract's keys plus
  roofline     -- algorithmic flops of one launch / HIP-event time of the kernel, vs the f32 MFMA peak of gfx950
                  (157.3 TFLOP/s; MI355X_MICROARCH.md)
  cpu_baseline -- the NumPy restatement of revrand's path (oracle/, kind "port") timed on this box's host cores over a
                  bounded row sample (rank 0, N=1 only), SURVEY 8d's recipe: 200 000 rows, median of 3
  configs      -- BASELINE.json's other configurations on this GPU (N=1 only): C2, C3 (one GPU's share), C4, C5 and the
                  headline shape in f64 arithmetic, each with its own roofline fraction and a bounded cpu sample.
"""
import argparse
import faulthandler
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X_MICROARCH.md "Peak FP64 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA", dense (same rate for fp16)
PEAK_HBM_TBS = 8.0            # MI355X_MICROARCH.md HBM3E

# HBM-side traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (tools/prof.sh),
# corrected as MI355X_MICROARCH.md prescribes; the committed summary is quoted, per launch.
TRAFFIC = {}
try:
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as _f:
        TRAFFIC = json.load(_f)
except Exception:
    pass


def posterior_flops(F):
    """Algorithmic flops of the posterior (slm.py:150-157: solve_posdef(iC, I) + m) as rr_posterior_dev executes it
    (rr_posdef.hip:5-10): potrf F^3/3 + inverse of the triangular factor F^3/3 + triangular Y^T Y F^3/3 = F^3 -- LAPACK's
    potrf + potri count.  (Rounds 2-5 quoted F^3/3 + F^3, a full-square inverse the kernels never form.)"""
    return float(F) ** 3


def flops_per_row(d, n, F=None):
    F = 2 * n if F is None else F
    return 2.0 * d * n + F * (F + 1.0) + 2.0 * F  # projection + upper-tri Gram + Phi^T y


def gen_chunk(c, rows, d, wvec):
    """Chunk c of the synthetic data set: X ~ N(0,1) f32, y = sin(X w) + 0.1 eps."""
    rng = np.random.default_rng([20260928, c])
    X = rng.standard_normal((rows, d), dtype=np.float32)
    y = np.sin(X @ wvec) + 0.1 * rng.standard_normal(rows, dtype=np.float32)
    return X, y.astype(np.float32)


def runtime_info(_hip, parallel):
    """Which HIP runtime and which librccl this process runs on (recorded in every line)."""
    info = {"hip_runtime": _hip.hip_runtime_path(), "RR_HIP_RUNTIME": os.environ.get("RR_HIP_RUNTIME", "system (default)")}
    try:
        ver, path = parallel.RcclComm.load()
        info["rccl"] = {"version": ver, "library": path}
    except Exception as e:  # N = 1 needs no RCCL: say so instead of failing the line
        info["rccl"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return info


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import revrand_oracle as orc  # checker/baseline only -- never on the product path
    return orc


def _blas_threads():
    threads = os.cpu_count()
    info = None
    try:
        from threadpoolctl import threadpool_info
        info = [{k: p.get(k) for k in ("user_api", "internal_api", "num_threads", "version")} for p in threadpool_info()]
        nt = [p.get("num_threads", 0) for p in info if p.get("user_api") == "blas"]
        threads = max(nt) if nt else threads
    except Exception:
        pass
    return int(threads), info


# ----------------------------------------------------------------------------------------------------
# the ONE JSON line: numbers only, under 8 KB (the driver keeps a 9 KB tail of stdout); prose lives in DESIGN.md 6
# ----------------------------------------------------------------------------------------------------

def _sig(x, digits=5):
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


def lean(obj):
    """What goes on stdout: keys starting with "_" (and everything below them) dropped, floats to 5 significant digits.
    The unabridged record -- every per-kernel time, sample description and note -- goes to --full-json."""
    if isinstance(obj, dict):
        return {k: lean(v) for k, v in obj.items() if not k.startswith("_")}
    if isinstance(obj, (list, tuple)):
        return [lean(v) for v in obj]
    if isinstance(obj, (np.floating,)):
        return _sig(float(obj))
    if isinstance(obj, (np.integer,)):
        return int(obj)
    return _sig(obj)


def full(obj):
    """The unabridged record: the same tree with the "_" prefixes removed."""
    if isinstance(obj, dict):
        return {k.lstrip("_"): full(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [full(v) for v in obj]
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (np.integer,)):
        return int(obj)
    return obj


class ParityError(AssertionError):
    pass


def parity(what, err, tol):
    """A side configuration whose result is off is not a measurement: it becomes {"error": ...} in the line (the headline
    asserts its own)."""
    if not (err <= tol):
        raise ParityError("parity: %s = %.3g exceeds its tolerance %.1g" % (what, err, tol))
    return float(err)


def cpu_baseline(d, n, W, wvec, sample_rows, budget_s=100.0):
    """the oracle's chunk-accumulated f64 path (10 000-row chunks) over a 200 000-row sample, wall-clock median
    of 3 (fewer when the budget is exhausted -- said in `sample`), BLAS on all host cores."""
    orc = _oracle()
    X, y = gen_chunk(10 ** 6, sample_rows, d, wvec)
    X64, y64 = X.astype(np.float64), y.astype(np.float64)
    orc.rff_gram_chunked(X64[:2000], y64[:2000], W, 1.0, chunk=1000)  # warm BLAS
    times, t_all = [], time.perf_counter()
    for _ in range(3):
        t0 = time.perf_counter()
        orc.rff_gram_chunked(X64, y64, W, 1.0, chunk=10000)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all + times[-1] > budget_s:
            break
    dt = float(np.median(times))
    threads, info = _blas_threads()
    sys.stderr.write("cpu_baseline: os.cpu_count()=%s threadpool_info=%s\n" % (os.cpu_count(), json.dumps(info)))
    return {"value": sample_rows / dt, "unit": "feature-rows/s", "cores": threads, "kind": "port",
            "sample": "%d rows of the same workload, f64, 10000-row chunks, median of %d runs (%s s)" % (
                sample_rows, len(times), ", ".join("%.1f" % t for t in times)),
            "os_cpu_count": os.cpu_count()}


# ----------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a launcher's environment starts its own N ranks
# ----------------------------------------------------------------------------------------------------

def launch(args, argv):
    """Start one rank per GPU (rank r on device r % visible GPUs), wait for all of them; a rank that fails -- or the whole
    job exceeding --launch-timeout -- ends every rank, and the launcher exits non-zero with the tail of the failing
    rank's stderr.  Rank 0's stdout (the ONE JSON line) passes through; every rank's stderr is forwarded."""
    from revrand_amd import _hip
    n = _hip.ctypes.c_int()
    lib = _hip.load_library()
    ndev = n.value if lib.rr_device_count(_hip.ctypes.byref(n)) == 0 else 0
    if ndev <= 0:
        raise SystemExit("bench.py: no HIP device visible (the hot path has no CPU fallback)")
    world = args.gpus
    rdzv_dir = tempfile.mkdtemp(prefix="rr_bench_")
    procs, logs = [], []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_RANK": str(r % ndev), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "RR_COMM_RDZV": "file:" + os.path.join(rdzv_dir, "rccl.id"),
                    "RR_BENCH_VISIBLE_GPUS": str(ndev)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world > ndev:
            # fewer GPUs than ranks (plumbing runs on a 1-GPU box): RCCL refuses two ranks on one device of one host,
            # so every rank claims its own host id and the exchange takes RCCL's socket transport
            env["NCCL_HOSTID"] = "rr-bench-rank-%d" % r
        logs.append(open(os.path.join(rdzv_dir, "rank%d.err" % r), "w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stderr=logs[-1]))
    rc, failed, why = 0, None, ""
    deadline = time.time() + args.launch_timeout
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc, failed, why = code, procs.index(p), "exited with status %d" % code
                    for q in pending:  # a rank failed: its peers would wait in the collective forever
                        q.terminate()
            if pending and rc == 0 and time.time() > deadline:
                rc, failed = 124, procs.index(pending[0])
                why = "still running after --launch-timeout %.0f s (%d of %d ranks unfinished)" % (
                    args.launch_timeout, len(pending), world)
                for q in pending:
                    q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        for r, f in enumerate(logs):
            f.flush()
            f.seek(0)
            text = f.read()
            f.close()
            if text.strip():
                sys.stderr.write("".join("[rank %d] %s\n" % (r, l) for l in text.splitlines()[-400:]))
            if failed == r:
                sys.stderr.write("bench.py: rank %d %s; its last output:\n%s\n" % (r, why, text[-3000:]))
        try:
            for f in os.listdir(rdzv_dir):
                os.unlink(os.path.join(rdzv_dir, f))
            os.rmdir(rdzv_dir)
        except OSError:
            pass
    raise SystemExit(rc)


# ----------------------------------------------------------------------------------------------------
# BASELINE.json's other configurations on this GPU (N = 1 only; never `value`)
# ----------------------------------------------------------------------------------------------------

def _timed(dev, fn, reps):
    """Median wall-clock of fn() with the device idle before and after (ms)."""
    ts = []
    for _ in range(reps):
        dev.sync()
        t0 = time.perf_counter()
        fn()
        dev.sync()
        ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts))


def _cpu(value, unit, sample):
    return {"value": value, "unit": unit, "cores": _blas_threads()[0], "kind": "port", "_sample": sample}


def _gram_step(dev, _hip, basis, dX, dy, F):
    """(step(), kernel timings list, stat pointers, accumulator) of one resident Gram pass: zero, features + SYRK, mirror."""
    acc = dev.zeros((F * F + F + 1) * 8)
    p = [_hip.ctypes.c_void_p(acc.ptr.value + o * 8) for o in (0, F * F, F * F + F)]
    kms = []

    def step():
        dev.memset(acc)
        basis.gram_dev(dX, dy, 1.0, *p)
        kms.append(basis.gram_timings())
        basis.symmetrize_dev(p[0])
    return step, kms, p, acc


def _gram_slice_parity(dev, _hip, basis, dX, dy, acc, p, F, X64, y64, W, ns, dtype):
    """The Gram of the first `ns` resident rows against the oracle's on the same rows (max-norm relative error)."""
    orc = _oracle()
    dXs = _hip.DeviceMatrix(dev, _hip.ctypes.c_void_p(dX.ptr.value), (ns, X64.shape[1]), dX.ld, dtype)
    dev.memset(acc)
    basis.gram_dev(dXs, _hip.DeviceView(dy, 0, ns), 1.0, *p)
    basis.symmetrize_dev(p[0])
    dXs.ptr = None
    Gs = dev.download(acc, (F, F), np.float64)
    Gr, _, _ = orc.rff_gram_chunked(X64[:ns], y64[:ns], W, 1.0)
    return float(np.abs(Gs - Gr).max() / np.abs(Gr).max())


def config_c2(dev, _hip, args):
    """configs[1]: RandomRBF F=4096, D=32, N=1M fp32: Phi + Phi^T Phi (+ Phi^T y)."""
    d, n, N = 32, 2048, 1_000_000
    F = 2 * n
    W = np.random.RandomState(42).randn(d, n)
    wvec = np.random.RandomState(1).randn(d).astype(np.float32)
    basis = _hip.RffHandle(W, compute="f32")
    X, y = gen_chunk(77, N, d, wvec)
    dX, dy = basis.upload(X), dev.upload_vector(y)
    step, kms, p, acc = _gram_step(dev, _hip, basis, dX, dy, F)
    step()
    kms.clear()
    ms = _timed(dev, step, 3)
    syrk = float(np.mean([k[1] + k[2] for k in kms]))
    feat = float(np.mean([k[0] for k in kms]))
    G = dev.download(acc, (F, F), np.float64)
    trace_err = parity("trace(G)/N - 1", abs(float(np.trace(G)) - N) / N, 1e-6)
    del G
    perr = None
    if not args.no_parity_check:
        perr = parity("Gram of 4096 rows vs oracle", _gram_slice_parity(
            dev, _hip, basis, dX, dy, acc, p, F, X.astype(np.float64), y.astype(np.float64), W, 4096, np.float32), 1e-4)
    cpu = None
    if not args.no_cpu_baseline:
        orc = _oracle()
        ns = 40_000
        t0 = time.perf_counter()
        orc.rff_gram_chunked(X[:ns].astype(np.float64), y[:ns].astype(np.float64), W, 1.0, chunk=10000)
        tc = time.perf_counter() - t0
        cpu = _cpu(ns / tc, "rows/s", "%d rows, f64, 10000-row chunks, one run %.1f s" % (ns, tc))
    for b in (dX, dy, acc):
        b.free()
    fl = flops_per_row(d, n)
    return {"workload": "RandomRBF F=4096 D=32 N=1M f32 resident: features + Gram", "rows": N,
            "ms": ms, "value": N / (ms * 1e-3), "unit": "rows/s", "dtype": "f32",
            "_launches_per_pass": kms[0][3], "_rows_per_launch": N // kms[0][3],
            "parity": {"gram_4096_rows": perr, "trace": trace_err},
            "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "frac": fl * N / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "syrk_frac": F * (F + 1.0) * N / (syrk * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "_whole_path_achieved": fl * N / (ms * 1e-3) / 1e12, "_syrk_kernels_ms": syrk,
                         "_features_kernel_ms": feat},
            "cpu_baseline": cpu}


def config_f64(dev, _hip, args):
    """The headline shape in the reference's own arithmetic (float64 features, f64 MFMA Gram) -- at the metric's N = 10M when
    the run is the full-size one (--rows >= 10M: every row of BASELINE's metric through float64, 2 timed passes of ~2.5 s),
    else on 500 000 rows."""
    d, n = 32, 2048
    N = 10_000_000 if args.rows >= 10_000_000 else min(500_000, max(args.rows, 4096))
    F = 2 * n
    W = np.random.RandomState(42).randn(d, n)
    wvec = np.random.RandomState(1).randn(d).astype(np.float32)
    basis = _hip.RffHandle(W, compute="f64")
    CH = 1_000_000
    X64, y64 = np.empty((N, d), dtype=np.float64), np.empty(N, dtype=np.float64)
    for c in range((N + CH - 1) // CH):   # (the headline's own chunk streams when N is the headline's)
        r0, r1 = c * CH, min(N, (c + 1) * CH)
        Xc, yc = gen_chunk(c if N >= 10_000_000 else 78, r1 - r0, d, wvec)
        X64[r0:r1], y64[r0:r1] = Xc, yc
    dX, dy = basis.upload(X64), dev.upload_vector(y64)
    step, kms, p, acc = _gram_step(dev, _hip, basis, dX, dy, F)
    step()
    kms.clear()
    ms = _timed(dev, step, 2 if N >= 10_000_000 else 3)
    syrk = float(np.mean([k[1] + k[2] for k in kms]))
    feat = float(np.mean([k[0] for k in kms]))
    G = dev.download(acc, (F, F), np.float64)
    trace_err = parity("trace(G)/N - 1", abs(float(np.trace(G)) - N) / N, 1e-12)
    del G
    perr = parity("Gram of 4096 rows vs oracle",
                  _gram_slice_parity(dev, _hip, basis, dX, dy, acc, p, F, X64, y64, W, 4096, np.float64), 1e-10)
    for b in (dX, dy, acc):
        b.free()
    return {"workload": "RandomRBF F=4096 D=32 N=%d float64 end to end: features + f64 MFMA Gram" % N, "rows": N,
            "ms": ms, "value": N / (ms * 1e-3), "unit": "rows/s", "dtype": "f64",
            "_launches_per_pass": kms[0][3], "_rows_per_launch": N // kms[0][3],
            "parity": {"gram_4096_rows": perr, "trace": trace_err},
            "roofline": {"bound": "mfma", "peak": PEAK_F64_MFMA_TFLOPS,
                         "frac": flops_per_row(d, n) * N / (ms * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                         "syrk_frac": F * (F + 1.0) * N / (syrk * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                         "_kernel": basis.gram_kernel_name(), "_syrk_kernels_ms": syrk, "_features_kernel_ms": feat}}


def config_laplace(dev, _hip, args):
    """RandomLaplace (Cauchy frequencies, |W| up to ~1e5: phases of ~1e5 revolutions) at config 2's shape, in its default
    arithmetic: the f32 pipeline behind float64 phases (RR_F32P64) -- X resident in float64, projection on the f64 matrix
    cores, phase reduced in float64, float32 sin / cos, then the same f32 MFMA Gram as RandomRBF."""
    import revrand_amd.basis_functions as bs
    d, n, N = 32, 2048, 1_000_000
    F = 2 * n
    wvec = np.random.RandomState(1).randn(d).astype(np.float32)
    X, y = gen_chunk(79, N, d, wvec)
    X64 = X.astype(np.float64) * (1.0 + 2.0 ** -30)  # not float32-representable, like real float64 data
    y64 = y.astype(np.float64)
    lap = bs.RandomLaplace(nbases=n, Xdim=d, random_state=42)
    assert lap.dtype == "f32" and lap.phase64
    basis = lap._handle()
    dX, dy = basis.upload(X64), dev.upload_vector(y64)
    step, kms, p, acc = _gram_step(dev, _hip, basis, dX, dy, F)
    step()
    kms.clear()
    ms = _timed(dev, step, 3)
    syrk = float(np.mean([k[1] + k[2] for k in kms]))
    feat = float(np.mean([k[0] for k in kms]))
    G = dev.download(acc, (F, F), np.float64)
    trace_err = parity("trace(G)/N - 1", abs(float(np.trace(G)) - N) / N, 1e-5)
    del G
    perr = parity("Gram of 4096 rows vs oracle",
                  _gram_slice_parity(dev, _hip, basis, dX, dy, acc, p, F, X64, y64, lap.W, 4096, np.float64), 1e-4)
    for b in (dX, dy, acc):
        b.free()
    fl = flops_per_row(d, n)
    return {"workload": "RandomLaplace F=4096 D=32 N=1M (max|W| %.2g): f64 phases, f32 features + Gram" % float(np.abs(lap.W).max()),
            "rows": N, "ms": ms, "value": N / (ms * 1e-3), "unit": "rows/s", "dtype": "f32 (f64 phases)",
            "parity": {"gram_4096_rows": perr, "trace": trace_err},
            "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "frac": fl * N / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "features_write_TBps": 4.0 * F * N / (feat * 1e-3) / 1e12,
                         "_syrk_kernels_ms": syrk, "_features_kernel_ms": feat,
                         "_features_kernel": "rr_rff_features_mfma64_kernel<32, 4, true, double, float>"}}


def _c3_data(rows, d, stream):
    """Rows of config 3's synthetic data set (chunk `stream` of it): X ~ N(0,1), y = sin(X w / sqrt(d)) + 0.1 eps."""
    w = np.random.default_rng([20260928, 3]).standard_normal(d, dtype=np.float32)
    rng = np.random.default_rng([20260928, 3, stream])
    X = rng.standard_normal((rows, d), dtype=np.float32)
    y = (np.sin(X @ w / np.sqrt(d)) + 0.1 * rng.standard_normal(rows, dtype=np.float32)).astype(np.float32)
    return X, y


def _c3_basis(d=64, n=4096):
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    return bs.RandomMatern52(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) \
        + bs.LinearBasis(onescol=True)


def config_c3(dev, _hip, args):
    """configs[2], one GPU's share: RandomMatern52 n=4096 + LinearBasis, D=64, N = 10M / 8 rows."""
    d, n, N = 64, 4096, 1_250_000
    X, y = _c3_data(N, d, 0)
    cat = _c3_basis(d, n)
    st = cat.device_fit_state(X, y)
    F = st.F
    hyp = [np.ones(d)]
    chunks = [rows for _, rows in st._chunks()]
    st.gram_device(hyp)
    ms = _timed(dev, lambda: st.gram_device(hyp), 3)
    # size-independent property on the full-size result: trace of the random Fourier block == N
    G, b, yty = st.stats_host()
    tr = parity("trace of the Fourier block / N - 1", abs(float(np.trace(G[:2 * n, :2 * n])) - N) / N, 1e-5)
    assert G[2 * n, 2 * n] == N and np.array_equal(G, G.T)
    del G
    # the rest of one `_elbo` at this shape (slm.py:154-199): posterior of the F_tot x F_tot system in HBM, second pass
    elbo = None
    if _hip.posterior_available(F):
        iL, var = np.full(F, 1.0), 0.5
        st.posterior(iL, var)
        t_post, post = _median_ms(lambda: st.posterior(iL, var), 2)
        st.second_pass(hyp, post[0], st.dC, var)
        t_p2, _ = _median_ms(lambda: st.second_pass(hyp, post[0], st.dC, var), 2)
        fl_post, fl_p2 = posterior_flops(F), 2.0 * F * F + 4.0 * d * n
        elbo = {"ms": {"statistics": ms, "posterior": t_post, "second_pass": t_p2},
                "posterior_frac_f64": fl_post / (t_post * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                "second_pass_frac": fl_p2 * N / (t_p2 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "frac": (2.0 * d * n + F * (F + 1.0) + 2.0 * F + fl_p2) * N / ((ms + t_post + t_p2) * 1e-3) / 1e12
                / PEAK_F32_MFMA_TFLOPS}
    st.release()
    cpu = None
    if not args.no_cpu_baseline:
        orc = _oracle()
        ns = 8000
        Xs, ys = X[:ns].astype(np.float64), y[:ns].astype(np.float64)
        Wm = cat.bases[0].W
        t0 = time.perf_counter()
        Phi = np.hstack((orc.rff_transform(Xs, Wm, np.ones(d)), orc.linear_transform(Xs, True)))
        orc.gram_stats(Phi, ys)
        tc = time.perf_counter() - t0
        cpu = _cpu(ns / tc, "rows/s", "%d rows, f64 transform + hstack + Phi^T Phi, one run %.1f s" % (ns, tc))
    fl = 2.0 * d * n + F * (F + 1.0) + 2.0 * F
    return {"workload": "RandomMatern52 n=4096 + LinearBasis D=64 F=%d N=1.25M (1/8 of 10M): concat + Gram, resident" % F,
            "rows": N, "ms": ms, "value": N / (ms * 1e-3), "unit": "rows/s", "dtype": "f32",
            "parity": {"trace_fourier_block": tr},
            "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "frac": fl * N / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "_flops_per_row": fl},
            "exchange_bytes": 8 * (F * (F + 1) // 2 + F + 2), "elbo_eval": elbo,
            "rows_per_launch": chunks, "cpu_baseline": cpu}


def config_c4(dev, _hip, args):
    """configs[3]: FastFoodRBF nbases=8192, D=128 (F=16384): the Hadamard / permute / diagonal chain, Phi streamed in
    262 144-row chunks into a two-slot device ring (Phi for N=4M is 262 GB in f32)."""
    import revrand_amd.basis_functions as bs
    d, nb, CH, NCH = 128, 8192, 262_144, 16  # N = 4 194 304 = BASELINE's 4M
    f = bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=1)
    h = f._handles()[0]  # the chain kernel's handle (rr_fastfood_*)
    F = 2 * h.n
    rng = np.random.default_rng([20260928, 4])
    X = rng.standard_normal((CH * NCH, d), dtype=np.float32)  # 2 GiB, resident before anything is timed
    dX = dev.upload_matrix(X)
    ring = [dev.malloc(CH * F * 4) for _ in range(2)]

    def pass_():
        for c in range(NCH):
            v = _hip.DeviceView(dX, c * CH, CH)
            h.transform_dev(v, 1.0, ring[c & 1], np.float32)
    pass_()
    dev.sync()
    dev.timer_start()
    reps = 2
    for _ in range(reps):
        pass_()
    kms = dev.timer_stop() / reps
    N = CH * NCH
    # parity of the last chunk's first rows against the oracle chain
    orc = _oracle()
    ns = 256
    out = dev.download(ring[(NCH - 1) & 1], (ns, F), np.float32)
    ref = orc.fastfood_transform(X[(NCH - 1) * CH:(NCH - 1) * CH + ns].astype(np.float64), f.B, f.G, f.PI, f.S, 1.0)
    perr = parity("Phi of 256 rows vs oracle chain", float(np.abs(out - ref).max() / np.abs(ref).max()), 1e-3)
    # unit row norm: sum_j Phi_j^2 = 1 for every row (cos^2 + sin^2 over n frequencies, / n)
    nrm = parity("row norm - 1", float(np.abs((out.astype(np.float64) ** 2).sum(axis=1) - 1.0).max()), 1e-4)
    cpu = None
    if not args.no_cpu_baseline:
        t0 = time.perf_counter()
        nc = 2000
        orc.fastfood_transform(X[:nc].astype(np.float64), f.B, f.G, f.PI, f.S, 1.0)
        tc = time.perf_counter() - t0
        cpu = _cpu(nc / tc, "rows/s", "%d rows through the oracle's NumPy FWHT chain, %.1f s" % (nc, tc))
    dX.free()
    for r in ring:
        r.free()
    bytes_row = 4.0 * d + 4.0 * F
    return {"workload": "FastFoodRBF nbases=8192 D=128 F=%d N=%d resident, %d chunks of %d into a 2-slot ring, f32" % (F, N, NCH, CH),
            "rows": N, "ms": kms, "value": N / (kms * 1e-3), "unit": "rows/s", "dtype": "f32",
            "parity": {"phi_256_rows": perr, "row_norm": nrm},
            "roofline": {"bound": "hbm", "kernel": "rr_fastfood16_kernel", "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                         "achieved": bytes_row * N / (kms * 1e-3) / 1e9, "frac": bytes_row * N / (kms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                         "bytes_per_row": bytes_row, "avg_launch_ms": kms / NCH, "_rows_per_launch": CH},
            "cpu_baseline": cpu}


def config_c4gm(dev, _hip, args, N_elbo=131_072):
    """FastFoodGM (one Gaussian spectral-mixture component, basis_functions.py:1386-1562) at config 4's width: nbases = 4096,
    D = 128 -> n = 4096, F = 4 n = 16384.  (a) the features by the chain kernel's mixture mode, streamed like config 4
    (HBM-write bound: 4 D + 4 F bytes per row); (b) one resident `_elbo`: chain -> MFMA SYRK at F = 16384 -> posterior ->
    second pass contracting BOTH gradients (mean, lenscale) on the device, against the oracle chain on 64 rows."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Bound, Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    orc = _oracle()
    d, nb, CH, NCH = 128, 4096, 262_144, 8

    def make():
        return bs.FastFoodGM(nbases=nb, Xdim=d, random_state=1, mean=Parameter(np.zeros(d), Bound()),
                             lenscale=Parameter(np.ones(d), Positive()))
    f = make()
    h = f._handles()[0]
    F = 4 * h.n
    rng = np.random.default_rng([20260928, 44])
    X = rng.standard_normal((CH * NCH, d), dtype=np.float32)
    mean, ls = 0.3 * np.random.RandomState(2).randn(d), np.linspace(0.8, 1.3, d)
    dX = dev.upload_matrix(X)
    ring = [dev.malloc(CH * F * 4) for _ in range(2)]

    def pass_():
        for c in range(NCH):
            h.gm_transform_dev(_hip.DeviceView(dX, c * CH, CH), mean, ls, ring[c & 1], np.float32)
    pass_()
    dev.sync()
    dev.timer_start()
    reps = 2
    for _ in range(reps):
        pass_()
    kms = dev.timer_stop() / reps
    N = CH * NCH
    ns = 256
    out = dev.download(ring[(NCH - 1) & 1], (ns, F), np.float32)
    ref = orc.fastfood_gm_transform(X[(NCH - 1) * CH:(NCH - 1) * CH + ns].astype(np.float64), f.B, f.G, f.PI, f.S, mean, ls)
    perr = parity("Phi of 256 rows vs oracle chain", float(np.abs(out - ref).max() / np.abs(ref).max()), 1e-3)
    nrm = parity("row norm - 1", float(np.abs((out.astype(np.float64) ** 2).sum(axis=1) - 1.0).max()), 1e-4)
    dX.free()
    for r in ring:
        r.free()
    # (b) one resident _elbo
    Xe = X[:N_elbo]
    ye = (np.sin(Xe @ np.random.RandomState(1).randn(d).astype(np.float32) / np.sqrt(d)) + 0.1 * rng.standard_normal(N_elbo, dtype=np.float32)).astype(np.float32)
    var, reg = 0.5, 1.0
    slm = StandardLinearModel(f)
    slm.obj_ = -np.inf
    slm._defer_cov = True
    st = slm._state = slm._make_state(Xe, ye)
    assert type(st).__name__ == "CatFitState"
    slm._elbo(Xe, ye, var, reg, [mean, ls])  # warm
    t_eval, res = _median_ms(lambda: slm._elbo(Xe, ye, var, reg, [mean, ls * 1.0]), reps=2)
    st.release()
    slm._state = None
    e_err = g_err = None
    if not args.no_parity_check:
        rows = 64
        Xs, ys = np.ascontiguousarray(Xe[:rows]), np.ascontiguousarray(ye[:rows])
        one = StandardLinearModel(make())
        one.obj_ = -np.inf
        one._state = one._make_state(Xs, ys)
        fn, (gv, gr, gh) = one._elbo(Xs, ys, var, reg, [mean, ls])
        one._state.release()
        one._state = None
        X64 = Xs.astype(np.float64)
        Phi = orc.fastfood_gm_transform(X64, f.B, f.G, f.PI, f.S, mean, ls)
        dM, dL = orc.fastfood_gm_grad(X64, f.B, f.G, f.PI, f.S, mean, ls)
        slabs = [np.ascontiguousarray(dM[:, :, i]) for i in range(d)] + [np.ascontiguousarray(dL[:, :, i]) for i in range(d)]
        o = orc.slm_elbo(Phi, ys.astype(np.float64), var, np.full(F, reg), slice(None), slabs)
        got = np.concatenate(([gv], np.atleast_1d(gr), gh[0], gh[1]))
        want = np.concatenate(([-o["dvar"]], [-g for g in o["dreg"]], [-g for g in o["dhyp"]]))
        e_err = parity("-ELBO of 64 rows vs oracle chain", abs(fn + o["elbo"]) / abs(o["elbo"]), 1e-4)
        g_err = parity("gradient of 64 rows vs oracle chain (normwise)", float(np.linalg.norm(got - want) / np.linalg.norm(want)), 2e-3)
    bytes_row = 4.0 * d + 4.0 * F
    fl_row = F * (F + 1.0) + 2.0 * F + 2.0 * F * F + 4.0 * d * F / 2  # Gram + Phi^T y + U = Phi C + the two (d, n) contractions of each half
    return {"workload": "FastFoodGM nbases=4096 D=128 F=%d: chain features N=%d in %d chunks; resident _elbo N=%d" % (F, N, NCH, N_elbo),
            "rows": N, "ms": kms, "value": N / (kms * 1e-3), "unit": "rows/s", "dtype": "f32",
            "elbo": {"rows": N_elbo, "ms": t_eval, "_neg_elbo": float(res[0]),
                     "frac_f32_mfma": fl_row * N_elbo / (t_eval * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
            "parity": {"phi_256_rows": perr, "row_norm": nrm, "neg_elbo_64_rows": e_err, "gradient_64_rows": g_err},
            "roofline": {"bound": "hbm", "kernel": "rr_fastfood16_kernel<..., GM>", "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                         "achieved": bytes_row * N / (kms * 1e-3) / 1e9, "frac": bytes_row * N / (kms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                         "bytes_per_row": bytes_row, "avg_launch_ms": kms / NCH, "_rows_per_launch": CH}}


def _ff_elbo_parity(make_basis, X, y, var, reg, ls, rows=64, nh=8):
    """One `_elbo` of the product on a FastFoodRBF basis (resident route, chain kernel, ONE process) on the first `rows` rows
    against the oracle's FWHT chain in float64: -ELBO, dvar, dreg and the first `nh` of the length-scale gradients (each is an
    F x rows x F product on the host).  (rel. error of -ELBO, normwise error of the gradient)."""
    from revrand_amd.slm import StandardLinearModel
    orc = _oracle()
    Xs, ys = np.ascontiguousarray(X[:rows]), np.ascontiguousarray(y[:rows])
    fb = make_basis()
    F = 2 * fb.n
    s2 = StandardLinearModel(fb)
    s2.obj_ = -np.inf
    s2._state = fb.device_fit_state(Xs, ys)
    nel, (gv, gr, gh) = s2._elbo(Xs, ys, var, reg, ls)
    s2._state.release()
    s2._state = None
    X64 = Xs.astype(np.float64)
    Phi = orc.fastfood_transform(X64, fb.B, fb.G, fb.PI, fb.S, ls)
    dP = orc.fastfood_grad(X64, fb.B, fb.G, fb.PI, fb.S, ls)
    ref = orc.slm_elbo(Phi, ys.astype(np.float64), var, np.full(F, reg), slice(None), [dP[:, :, i] for i in range(nh)])
    del dP
    got = np.concatenate(([gv], np.atleast_1d(gr), np.atleast_1d(gh)[:nh]))
    want = np.concatenate(([-ref["dvar"]], [-g for g in ref["dreg"]], [-g for g in ref["dhyp"]]))
    return abs(nel + ref["elbo"]) / abs(ref["elbo"]), float(np.linalg.norm(got - want) / np.linalg.norm(want))


def config_ff_elbo(dev, _hip, args, N=524_288):
    """configs[3] WITH the Gram (SURVEY 8 a-11 "C4-with-Gram", f-4's width): one `_elbo` of StandardLinearModel on FastFoodRBF
    nbases=8192, D=128 ARD (F = 16384) over one GPU's share of N = 4M rows (4 194 304 / 8), resident: the statistics pass
    runs the chain kernel into the device feature matrix and the MFMA SYRK at F = 16384, the posterior is the 128-panel
    Cholesky + inverse in HBM, the second pass contracts X^T A against the same features (basis_functions.py:1263-1289,
    slm.py:142-199)."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    d, nb = 128, 8192
    rng = np.random.default_rng([20260928, 44])
    X = rng.standard_normal((N, d), dtype=np.float32)
    w = rng.standard_normal(d, dtype=np.float32)
    y = (np.sin(X @ w / np.sqrt(d)) + 0.1 * rng.standard_normal(N, dtype=np.float32)).astype(np.float32)

    def make_basis():
        return bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
    f = make_basis()
    F = 2 * f.n
    slm = StandardLinearModel(f)
    slm.obj_ = -np.inf
    slm._defer_cov = True
    st = slm._state = f.device_fit_state(X, y)
    assert type(st).__name__ == "CatFitState" and _hip.posterior_available(F)
    ls, var, reg = np.linspace(0.8, 1.3, d), 0.5, 1.0
    iL = np.full(F, 1.0 / reg)
    slm._elbo(X, y, var, reg, ls)  # warm: scratch allocations, posterior work space
    t_stats, _ = _median_ms(lambda: st.gram_device([ls]), 2)
    G, _, _ = st.stats_host()
    tr = parity("trace(G)/N - 1", abs(float(np.trace(G)) - N) / N, 1e-5)
    del G
    t_post, post = _median_ms(lambda: st.posterior(iL, var), 2)
    t_pass2, _ = _median_ms(lambda: st.second_pass([ls], post[0], st.dC, var), 2)
    t_eval, res = _median_ms(lambda: slm._elbo(X, y, var, reg, ls * 1.0), 2)
    chunks = [rows for _, rows in st._chunks()]
    st.release()
    slm._state = None
    perr = (None, None)
    if not args.no_parity_check:
        e0, e1 = _ff_elbo_parity(make_basis, X, y, var, reg, ls)
        perr = (parity("-ELBO of 64 rows vs the oracle's chain", e0, 1e-5),
                parity("gradient of 64 rows vs the oracle's chain (normwise)", e1, 1e-3))
    fl_stats = F * (F + 1.0) + 2.0 * F + f.k * (2.0 * f.d2 * np.log2(f.d2) + 3.0 * f.d2)
    fl_pass2 = 2.0 * F * F + 2.0 * d * F
    fl_post = posterior_flops(F)
    return {"workload": "StandardLinearModel._elbo, FastFoodRBF nbases=8192 D=128 ARD (F=%d), N=%d (1/8 of 4M) resident, f32" % (F, N),
            "rows": N, "dtype": "f32", "ms": t_eval, "value": N / (t_eval * 1e-3), "unit": "rows/s per _elbo",
            "stage_ms": {"statistics": t_stats, "posterior": t_post, "second_pass": t_pass2}, "_rows_per_launch": chunks,
            "_neg_elbo": float(res[0]),
            "parity": {"trace": tr, "neg_elbo_64_rows": perr[0], "gradient_64_rows": perr[1]},
            "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS,
                         "frac": (fl_stats + fl_pass2) * N / (t_eval * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "statistics_frac": fl_stats * N / (t_stats * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "second_pass_frac": fl_pass2 * N / (t_pass2 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "posterior_frac_f64": fl_post / (t_post * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS}}


def config_glm_default(dev, _hip, args):
    """The reference's OWN GeneralizedLinearModel regime -- its defaults K = 10, nsamples = 50, batch_size = 10, maxiter = 3000,
    nstarts = 500 (glm.py:120-124) on its model test's shape (tests/test_models.py:83-112: N = 600 rows, LinearBasis +
    RandomRBF(20) + RandomMatern52(20), Gaussian likelihood): one `fit` -- 500 random starts + 3000 SGD steps of ~1.7 MFLOP
    each, where launches, not arithmetic, were all of the time -- through rr_glm_svi (many steps per kernel launch, the random
    starts as one launch).  Both samplers: "host" = the reference's random stream (the host generates 41 500 normals per
    evaluation: that is its bound), "device" = the counter-based generator.  Parity: the reference's own fit of
    tests/golden/glm_fit.npz (`gaussian_cat_bs10_ns5`: same three-child concatenation, batch 10, 5 starts, 20 steps) through
    the same loop, every fitted block and the stream's end state."""
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd.glm import GeneralizedLinearModel
    rs = np.random.RandomState(100)
    N = 600
    x = np.linspace(-5, 5, N)
    y = 3 + 2 * x + rs.randn(N) * 1e-4
    X = np.column_stack((np.ones(N), x))

    def make(sampler, **kw):
        basis = bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=20, Xdim=2) + bs.RandomMatern52(nbases=20, Xdim=2)
        return GeneralizedLinearModel(lk.Gaussian(), basis, random_state=1, sampler=sampler, **kw)
    out = {}
    launches = []
    real_run = _hip.FusedSvi.run

    def timed_run(self, n, *a, **k):   # one extra fit with every launch synchronised: device time per step
        self.dev.sync()
        t0 = time.perf_counter()
        r = real_run(self, n, *a, **k)
        self.dev.sync()
        launches.append((n, time.perf_counter() - t0))
        return r
    for sampler in ("host", "device"):
        ts = []
        for rep in range(4):
            g = make(sampler)
            np.random.seed(0)
            t0 = time.perf_counter()
            g.fit(X, y)
            ts.append(time.perf_counter() - t0)
        Ey = g.predict(X[:50])
        smse = float(((Ey - y[:50]) ** 2).mean() / y.var())
        assert smse < 0.1, smse   # (the reference's own test asserts this, tests/test_models.py:95)
        t_fit = float(np.median(ts[1:]))
        out[sampler] = {"fit_s": t_fit, "_us_per_evaluation": 1e6 * t_fit / 3500.0, "_fits_s": ts, "_smse": smse}
    _hip.FusedSvi.run = timed_run
    try:
        g = make("device")
        np.random.seed(0)
        g.fit(X, y)
    finally:
        _hip.FusedSvi.run = real_run
    steps = sum(n for n, _ in launches)
    dev_us = 1e6 * sum(t for _, t in launches) / steps
    assert steps == 3000, steps
    # the step-per-call loop (round 5's route for this fit) on the same box, device sampler
    g = make("device")
    g._fused_sgd = False
    np.random.seed(0)
    g.fit(X, y)
    g = make("device")
    g._fused_sgd = False
    np.random.seed(0)
    t0 = time.perf_counter()
    g.fit(X, y)
    t_r5 = time.perf_counter() - t0
    perr = None
    if not args.no_parity_check:
        with np.load(os.path.join(ROOT, "tests", "golden", "glm_fit.npz")) as z:
            gz = {k: z[k] for k in z.files}
        tag = "gaussian_cat_bs10_ns5"
        Xg = gz["X"]
        basis = bs.LinearBasis(onescol=True) + bs.RandomRBF(nbases=int(gz["nbases"]), Xdim=Xg.shape[1], random_state=3) \
            + bs.RandomMatern52(nbases=int(gz["nbases"]), Xdim=Xg.shape[1], random_state=4)
        g = GeneralizedLinearModel(lk.Gaussian(), basis, K=int(gz["K"]), nsamples=int(gz["L"]), batch_size=10, maxiter=int(gz["maxiter"]),
                                   nstarts=5, random_state=int(gz["seed"]))
        np.random.seed(int(gz["global_seed"]))
        g.fit(Xg, gz["y_gaussian"])

        def nw(a, b):
            a, b = np.ravel(np.asarray(a, float)), np.ravel(np.asarray(b, float))
            return float(np.abs(a - b).max() / np.abs(b).max())
        flat = lambda v: np.concatenate([np.ravel(np.asarray(u, float)) for u in (v if isinstance(v, (list, tuple)) else [v])])  # noqa: E731
        err = max(nw(g.weights_, gz[tag + "_m"]), nw(g.covariance_, gz[tag + "_C"]), nw(flat(g.regularizer_), gz[tag + "_reg"]),
                  nw(flat(g.like_hypers_), gz[tag + "_lik"]), nw(flat(g.basis_hypers_), gz[tag + "_ls"]))
        perr = {"fit_vs_reference": parity("fitted blocks of the reference's fit (normwise max)", err, 1e-4),
                "stream_end_state_equal": bool(g.random_.randn() == float(gz[tag + "_end"]))}
        assert perr["stream_end_state_equal"]
    # the oracle's port of the same loop on this box's host: steps per second from 40 steps (structured_sgd o logtrick_sgd o sgd)
    orc = _oracle()
    from scipy.stats import gamma
    P = orc.ParamSpec
    W1 = bs.RandomRBF(nbases=20, Xdim=2, random_state=5).W
    W2 = bs.RandomMatern52(nbases=20, Xdim=2, random_state=6).W
    reg = lambda: P(dist=gamma(1.), positive=True)  # noqa: E731
    t0 = time.perf_counter()
    orc.glm_fit(X, y, "gaussian", [], [("linear", True), ("rff", W1, 1), ("rff", W2, 1)], [reg(), reg(), reg()],
                [P(dist=gamma(1.), positive=True)], [P(value=[]), P(dist=gamma(1.), positive=True), P(dist=gamma(1.), positive=True)],
                10, 50, 10, 40, 0, 1, 0)
    cpu_s = (time.perf_counter() - t0) / 40.0
    threads, _ = _blas_threads()
    F, K, L, M = 83, 10, 50, 10
    flops = K * (2.0 * 2.0 * L * M * F) + 2.0 * M * 2 * 20 * 2   # the two sample products per component + the projections
    return {"workload": "GeneralizedLinearModel.fit at the reference's defaults (K=10, nsamples=50, batch_size=10, maxiter=3000, "
                        "nstarts=500), Gaussian, Linear + RandomRBF(20) + RandomMatern52(20), N=600 D=2: F=83",
            "dtype": "f64", "ms": 1e3 * out["host"]["fit_s"], "value": 3500.0 / out["host"]["fit_s"],
            "unit": "_elbo evaluations/s (3000 steps + 500 starts per fit)", "host": out["host"], "device": out["device"],
            "device_us_per_step": dev_us, "_launches": len(launches), "fit_s_step_per_call_loop_device_sampler": t_r5,
            "parity": perr,
            "cpu_baseline": {"value": 1.0 / cpu_s, "unit": "_elbo evaluations/s", "cores": threads, "kind": "port",
                             "sample": "oracle.glm_fit, 40 SGD steps of the same model on the host"},
            "roofline": {"bound": "launch latency", "peak": PEAK_F64_MFMA_TFLOPS, "frac": flops / (dev_us * 1e-6) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                         "_what": "1.7 MFLOP per step against the f64 MFMA peak: the regime is latency, not arithmetic"}}


def config_c5(dev, _hip, args):
    """configs[4]: GLM Poisson, RandomRBF F=2048, D=32 ARD, N=2M resident, K=10, L=50, minibatch 65 536: one SVI
    minibatch `_elbo` (Phi + ELBO gradients incl. the length-scale gradient)."""
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    N, d, n, K, L, M = 2_000_000, 32, 1024, 10, 50, 65536
    F = 2 * n
    rng = np.random.default_rng([20260928, 5])
    X = rng.standard_normal((N, d), dtype=np.float32)
    y = rng.poisson(np.exp(0.3 * X[:, 0].astype(np.float64))).astype(np.float64)
    rs = np.random.RandomState(0)
    m = 0.1 * rs.randn(F, K)
    C = rs.gamma(2., 0.5, size=(F, K))
    ls = np.linspace(0.8, 1.5, d)
    gemm_flops = 3 * 2.0 * K * L * M * F
    # Sessions: host, device, host again (a session that comes first after a configuration that freed tens of GB runs
    # ~0.5 ms per step slower than the same session a few seconds later: where its buffers land, not what it computes --
    # tools/ab.sh c5order).  EVERY session is reported; each route's figure is the MEDIAN of its sessions.
    sessions = {}
    for sampler in os.environ.get("RR_BENCH_C5_ORDER", "host,device,host").split(","):
        basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()))
        glm = GeneralizedLinearModel(lk.Poisson(), basis, K=K, nsamples=L, batch_size=M, random_state=2, sampler=sampler)
        glm.B_, glm.D_ = N / M, F
        glm._GeneralizedLinearModel__it = 1  # a plain SGD iteration (no ELBO logging)
        feats = glm._features()
        glm._resident_fit = feats.make_resident(X)
        assert glm._resident_fit
        perm = rs.permutation(N)
        Xstub = np.empty((M, 0))

        def step(i):
            idx = perm[(i * M) % (N - M):(i * M) % (N - M) + M]
            return glm._elbo(m, C, 1.0, [], ls, Xstub, y[idx], idx)
        step(0)
        dev.sync()
        reps = 8
        t0 = time.perf_counter()
        for i in range(reps):
            step(i + 1)
        dev.sync()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        raw = {"elbo_step_ms": ms}
        # the device calls of one step alone, the same way for both samplers (feature assembly, the step's kernels incl. its
        # three MFMA GEMMs, the length-scale contraction, 0.6 MB back): wall-clock, so launch gaps and the small transfers
        # are inside.  Reference stream: the draws are device-resident before the step, as `fit`'s worker leaves them.
        idx = perm[:M]
        Edev = None
        if sampler == "host":
            E = rs.standard_normal((K * L, F)).astype(np.float32)
            Edev = dev.upload_vector(E.ravel())
            Edev.shape, Edev.dtype = E.shape, E.dtype

        def device_calls(i):
            feats.assemble_idx(idx, [ls])
            if sampler == "device":
                feats.glm_step_sampled(y[idx], None, lk.RR_LIK_POISSON_EXP, 0.0, m, C, K, L, 7, i)
            else:
                feats.glm_step_draws(y[idx], None, lk.RR_LIK_POISSON_EXP, 0.0, m, C, K, L, Edev)
            feats.glm_basis_grads(Xstub)
        # (steady state, as inside `fit`: the draws above took the host ~30 ms during which the GPU clocked down -- three
        # untimed calls bring it back before the timed ones)
        for i in range(3):
            device_calls(i)
        dev.sync()
        dreps = 16
        t0 = time.perf_counter()
        for i in range(dreps):
            device_calls(i + 3)
        dev.sync()
        dms = 1e3 * (time.perf_counter() - t0) / dreps
        if Edev is not None:
            Edev.free()
        raw["device_calls_ms"] = dms
        glm._resident_fit = False
        glm._release_features()
        # the same step as `fit` runs it -- minibatch t+1 (and, for the reference's stream, its draws) made on a worker
        # thread while step t is on the device, optimiser update included: two fits of different length, so that the
        # one-off upload of X drops out
        import logging
        logging.getLogger("revrand_amd").setLevel(logging.ERROR)
        # `fit` as the estimator runs it: the RESIDENT loop (rr_glm_sgd: parameters, updater state and gradient in HBM, a step
        # queued per library call, nothing read back) and, beside it, the host loop around `_elbo` it replaces (the same fit:
        # tests/test_gpu_resident_sgd.py).  Resident: the mean interval between the moments the loop queues its steps (the
        # queue is two deep, so it follows the device) over steps 8 .. 198 of a 200-step fit -- epoch boundaries included;
        # host loop: two fits of different length, so that the upload of X drops out.  (A first fit of each kind is not
        # timed: the one-off allocations of its rings and contexts are not a step's.)
        def one_fit(iters, resident):
            g2 = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1,
                                                                   lenscale=Parameter(np.ones(d), Positive())),
                                        K=K, nsamples=L, batch_size=M, maxiter=iters, nstarts=0, random_state=2, sampler=sampler)
            g2._resident_sgd = resident
            np.random.seed(20260930)  # (the start point is a draw from NumPy's global stream)
            t0 = time.perf_counter()
            g2.fit(X, y)
            return time.perf_counter() - t0, g2
        _, g8 = one_fit(8, True)
        raw["_fit8"] = [np.concatenate((g8.weights_.ravel(), g8.covariance_.ravel(), np.atleast_1d(g8.basis_hypers_)))]
        t200, g200 = one_fit(200, True)
        ck = g200.__dict__["_resident_clock"]
        dt = 1e3 * np.diff(ck[8:199])
        raw["fit_step_ms"], raw["fit_step_median_ms"], raw["fit_step_max_ms"] = float(dt.mean()), float(np.median(dt)), float(dt.max())
        raw["_fit_200_steps_s"] = t200
        tfit = {}
        for iters in (8, 8, 72):
            tfit[iters], g2 = one_fit(iters, False)
            if iters == 8:
                h8 = np.concatenate((g2.weights_.ravel(), g2.covariance_.ravel(), np.atleast_1d(g2.basis_hypers_)))
        raw["fit_step_host_loop_ms"] = 1e3 * (tfit[72] - tfit[8]) / 64
        raw["_fit8"].append(h8)
        a8, b8 = raw.pop("_fit8")
        raw["resident_vs_host_loop_8_steps"] = parity("C5 %s: parameters after 8 steps, resident loop vs host loop (normwise)" % sampler,
                                                      float(np.linalg.norm(a8 - b8) / np.linalg.norm(b8)), 1e-4)
        sessions.setdefault(sampler, []).append(raw)
    out = {}
    for sampler, runs in sessions.items():
        ms, dms, fms, hms, mms = (float(np.median([r[k] for r in runs])) for k in ("elbo_step_ms", "device_calls_ms", "fit_step_ms",
                                                                                  "fit_step_host_loop_ms", "fit_step_median_ms"))
        out[sampler] = {"fit_step_ms": fms, "_fit_step_median_ms": mms, "fit_step_host_loop_ms": hms, "device_calls_ms": dms, "_elbo_step_ms": ms,
                        "device_calls_frac": gemm_flops / (dms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                        "fit_step_frac": gemm_flops / (fms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                        "resident_vs_host_loop_8_steps": max(r["resident_vs_host_loop_8_steps"] for r in runs),
                        "_sessions_fit_dev_ms": [[r["fit_step_ms"], r["device_calls_ms"]] for r in runs], "_sessions": runs}
    cpu = None
    if not args.no_cpu_baseline:
        orc = _oracle()
        Mc = 1024
        Xc, yc = X[:Mc].astype(np.float64), y[:Mc]
        W = basis.W
        e = rs.randn(K, L, F)
        t0 = time.perf_counter()
        Phi = orc.rff_transform(Xc, W, ls)
        dP = orc.rff_grad(Xc, W, ls)
        orc.glm_elbo(m, C, np.ones(F), slice(None), "poisson_exp", [], (), Phi, [dP[:, :, i] for i in range(d)], yc, e, N / M)
        tc = time.perf_counter() - t0
        cpu = _cpu(Mc / tc, "minibatch-rows/s", "%d-row minibatch through the oracle (transform, (M, F, d) grad, glm_elbo), %.1f s" % (Mc, tc))
    dflt = out["host"]  # the estimator's default route: the reference's random stream
    return {"workload": "GLM Poisson, RandomRBF F=2048 D=32 ARD, N=2M resident, K=10 L=50, minibatch 65536: SVI step of fit()",
            "rows_per_step": M, "ms": dflt["fit_step_ms"], "value": M / (dflt["fit_step_ms"] * 1e-3), "unit": "minibatch-rows/s",
            "dtype": "f32", "samplers": {"host": out["host"], "device": out.get("device")},
            "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "frac": dflt["fit_step_frac"],
                         "frac_host_loop_device_calls": dflt["device_calls_frac"], "_gemm_flops_per_step": gemm_flops,
                         "_what": "the step's three (K L) x M x F GEMMs over the wall-clock of a whole step of fit() -- the resident "
                                  "loop (rr_glm_sgd), default route (sampler='host', the reference's random stream), median of its "
                                  "sessions; frac_host_loop_device_calls: the same flops over the device calls of one step of the "
                                  "host loop around _elbo (rounds 2-4's figure)"},
            "cpu_baseline": cpu}


def _median_ms(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts)), r


def _oracle_features(basis, X64, hyp):
    """(Phi, [dPhi slabs], reg slices helper) of a random Fourier basis or of `that + LinearBasis` through the oracle."""
    orc = _oracle()
    bases = getattr(basis, "bases", [basis])
    ls = np.asarray(hyp[0] if isinstance(hyp, list) else hyp, dtype=float)
    Phi_r = orc.rff_transform(X64, bases[0].W, ls)
    dP = orc.rff_grad(X64, bases[0].W, ls)
    blocks = [Phi_r] + [orc.linear_transform(X64, True) for _ in bases[1:]]
    Phi = np.hstack(blocks)
    slabs = []
    for i in range(dP.shape[2]):
        full_ = np.zeros_like(Phi)
        full_[:, :Phi_r.shape[1]] = dP[:, :, i]
        slabs.append(full_)
    ends = np.cumsum([0] + [b.shape[1] for b in blocks])
    return Phi, slabs, [slice(int(ends[i]), int(ends[i + 1])) for i in range(len(blocks))]


def _elbo_parity(make_basis, X, y, var, reg, hyp, rows=256, devices=None):
    """One `_elbo` of the product on the first `rows` rows (resident route, device posterior, ONE process) against the
    oracle's slm_elbo on the same rows in float64: (rel. error of -ELBO, normwise error of [dvar, dreg, dhyp])."""
    from revrand_amd import parallel
    from revrand_amd.slm import StandardLinearModel
    orc = _oracle()
    Xs, ys = np.ascontiguousarray(X[:rows]), np.ascontiguousarray(y[:rows])
    basis = make_basis()
    slm = StandardLinearModel(basis, devices=devices)
    slm.obj_ = -np.inf
    slm._state = slm._make_state(Xs, ys)
    f, (gv, gr, gh) = slm._elbo(Xs, ys, var, reg, hyp)
    slm._state.release()
    slm._state = None
    Phi, slabs, slices = _oracle_features(basis, Xs.astype(np.float64), hyp)
    regs = np.atleast_1d(np.asarray(reg, dtype=float))
    diag = np.concatenate([np.full(s.stop - s.start, regs[min(i, len(regs) - 1)]) for i, s in enumerate(slices)])
    ref = orc.slm_elbo(Phi, ys.astype(np.float64), var, diag, slices if len(slices) > 1 else slices[0], slabs)
    got = np.concatenate(([gv], np.atleast_1d(gr), np.atleast_1d(gh)))
    want = np.concatenate(([-ref["dvar"]], [-g for g in ref["dreg"]], [-g for g in ref["dhyp"]]))
    return abs(f + ref["elbo"]) / abs(ref["elbo"]), float(np.linalg.norm(got - want) / np.linalg.norm(want))


def config_c1(dev, _hip, args):
    """BASELINE configs[0] -- RandomRBF nbases = 256, D = 8, N = 10k through StandardLinearModel (slm.py:74-199) -- as a
    LATENCY line: what a GridSearchCV user waits for per `_elbo` and per `fit` in the launch-bound regime, next to the same
    call of the oracle port on this box's host cores (0.375 s per `_elbo`, 21.1 s per fit for the
    reference in the survey container).  Data and start values are those of tests/golden/fit_c1.npz (oracle/make_golden.py
    c1_data / gen_fit_c1); parity: one `_elbo` at the reference's fitted point against the reference's own evaluation."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    orc = _oracle()
    N, d, n = 10_000, 8, 256
    r = np.random.RandomState(11)
    X = r.randn(N, d)
    y = np.sin(X @ np.array([1.0, -0.7, 0.5, 0.3, -0.2, 0.9, -0.4, 0.1])) + 0.1 * r.randn(N)
    with np.load(os.path.join(ROOT, "tests", "golden", "fit_c1.npz")) as z:
        g = {k: z[k] for k in z.files}

    def make(dtype="f32"):
        return bs.RandomRBF(nbases=n, Xdim=d, random_state=41, lenscale=Parameter(2.0, Positive()),
                            regularizer=Parameter(10.0, Positive()), dtype=dtype)
    var, reg, hyp = float(g["c1_var_"]), float(g["c1_reg_"]), float(g["c1_hyp_"])
    out = {}
    for dtype in ("f32", "f64"):
        slm = StandardLinearModel(make(dtype))
        slm.obj_ = -np.inf
        slm._defer_cov = True
        slm._state = slm._make_state(X, y)
        f, grads = slm._elbo(X, y, var, reg, hyp)  # warm; and the parity evaluation
        tol = 1e-3 if dtype == "f32" else 1e-7
        parity("%s: -ELBO at the reference's fitted point" % dtype, abs(-f - float(g["c1_at_elbo"])) / abs(float(g["c1_at_elbo"])), tol * 0.1)
        parity("%s: posterior weights there" % dtype, float(np.abs(slm.weights_ - g["c1_at_m"]).max() / np.abs(g["c1_at_m"]).max()), tol)
        ts = []
        for k in range(30):
            t0 = time.perf_counter()
            slm._elbo(X, y, var, reg, hyp * (1.0 + 1e-6 * k))  # a new length scale every call, as the optimiser's
            ts.append(1e3 * (time.perf_counter() - t0))
        slm._state.release()
        slm._state = None
        t_fit = []
        nev = 0
        for _ in range(3):
            est = StandardLinearModel(make(dtype), var=Parameter(0.02, Positive()), nstarts=0, maxiter=20, random_state=0)
            calls = [0]
            inner = StandardLinearModel._elbo_resident

            def counted(self_, *a, **k):
                calls[0] += 1
                return inner(self_, *a, **k)
            StandardLinearModel._elbo_resident = counted
            try:
                t0 = time.perf_counter()
                est.fit(X, y)
                t_fit.append(time.perf_counter() - t0)
            finally:
                StandardLinearModel._elbo_resident = inner
            nev = calls[0]
        out[dtype] = {"elbo_ms": float(np.median(ts)), "_elbo_ms_min": float(np.min(ts)), "fit_s": float(np.median(t_fit)),
                      "_fit_elbo_evaluations": nev, "fit_obj": float(est.obj_)}
    # the same `_elbo` of the oracle port (features, Gram, Cholesky, gradients: NumPy / BLAS on the host cores)
    W = make().W
    t_cpu = []
    for _ in range(3):
        t0 = time.perf_counter()
        Phi = orc.rff_transform(X, W, hyp)
        dP = orc.rff_grad(X, W, hyp)
        orc.slm_elbo(Phi, y, var, np.full(2 * n, reg), slice(None), [dP])
        t_cpu.append(time.perf_counter() - t0)
    threads, _ = _blas_threads()
    cpu_ms = 1e3 * float(np.median(t_cpu))
    F = 2 * n
    fl = (flops_per_row(d, n) + 2.0 * F * F + 4.0 * d * n) * N + posterior_flops(F)
    return {"workload": "StandardLinearModel, RandomRBF nbases=256 D=8 N=10k (BASELINE configs[0]): one resident _elbo and "
                        "fit(nstarts=0, maxiter=20)", "rows": N, "dtype": "f32",
            "ms": out["f32"]["elbo_ms"], "value": 1e3 / out["f32"]["elbo_ms"], "unit": "_elbo evaluations/s",
            "f32": out["f32"], "f64": out["f64"],
            "cpu_baseline": {"value": 1e3 / cpu_ms, "unit": "_elbo evaluations/s", "ms": cpu_ms, "cores": threads, "kind": "port",
                             "sample": "the same evaluation (features, gradient tensor, Gram, Cholesky) by oracle/revrand_oracle.py, f64, "
                                       "median of 3; BASELINE.md section 3 quotes 375 ms for the reference in the survey container"},
            "parity": {"fit_obj_vs_reference_f64": abs(out["f64"]["fit_obj"] - float(g["c1_obj"])) / abs(float(g["c1_obj"]))},
            "roofline": {"bound": "launch latency", "peak": PEAK_F32_MFMA_TFLOPS,
                         "frac": fl / (out["f32"]["elbo_ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "_note": "0.11 GFLOP per evaluation: 0.7 us of the matrix pipe; the call is host + launch latency"}}


def config_elbo(dev, _hip, args, dtype="f32", N=1_000_000):
    """One L-BFGS evaluation of StandardLinearModel._elbo (slm.py:142-199) at config 2's shape with the data resident:
    statistics pass (features + Gram), posterior in HBM (blocked Cholesky + inverse + reductions), second pass (features,
    U = Phi C, residual, hyper-gradient contraction).  Stage times are host wall-clock around calls that return numbers
    (each ends with a synchronising copy), the whole evaluation is `_elbo` as the optimiser calls it."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    d, n = 32, 2048
    F = 2 * n
    wvec = np.random.RandomState(1).randn(d).astype(np.float32)
    X, y = gen_chunk(91, N, d, wvec)
    if dtype == "f64":
        X, y = X.astype(np.float64), y.astype(np.float64)

    def make_basis():
        return bs.RandomRBF(nbases=n, Xdim=d, random_state=42, lenscale=Parameter(np.ones(d), Positive()), dtype=dtype)
    basis = make_basis()
    slm = StandardLinearModel(basis)
    slm.obj_ = -np.inf
    slm._defer_cov = True  # as inside fit(): the best posterior covariance stays in HBM until the optimiser is done
    st = slm._state = basis.device_fit_state(X, y)
    ls, var, reg = np.linspace(0.8, 1.3, d), 0.5, 1.0
    iL = np.full(F, 1.0 / reg)
    assert _hip.posterior_available(F)
    slm._elbo(X, y, var, reg, ls)  # warm: scratch allocations, posterior work space
    t_stats, _ = _median_ms(lambda: st.gram_device(ls))
    kt = st.handle.gram_timings()
    t_post, post = _median_ms(lambda: st.posterior(iL, var))
    m = post[0]
    t_pass2, _ = _median_ms(lambda: st.second_pass(ls, m, st.dC, var))
    t_eval, res = _median_ms(lambda: slm._elbo(X, y, var, reg, ls * 1.0))
    st.release()
    slm._state = None
    peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_F64_MFMA_TFLOPS
    fl_stats = flops_per_row(d, n)                    # 2dn + F(F+1) + 2F
    fl_pass2 = 2.0 * F * F + 2.0 * d * n + 2.0 * d * n  # U = Phi C, the features again, the (d, n) contraction X^T A
    fl_post = posterior_flops(F)                   # Cholesky + inverse from the factor (f64 MFMA)
    fl_row = fl_stats + fl_pass2
    perr = (None, None)
    if not args.no_parity_check:
        perr = _elbo_parity(make_basis, X, y, var, reg, ls)
        tol = (1e-4, 2e-3) if dtype == "f32" else (1e-10, 1e-9)
        parity("-ELBO of 256 rows vs oracle", perr[0], tol[0])
        parity("gradient of 256 rows vs oracle (normwise)", perr[1], tol[1])
    return {"workload": "StandardLinearModel._elbo, RandomRBF F=4096 D=32 ARD, N=%d resident, %s" % (N, dtype), "rows": N,
            "dtype": dtype, "ms": t_eval, "value": N / (t_eval * 1e-3), "unit": "rows/s per _elbo",
            "stage_ms": {"statistics": t_stats, "posterior": t_post, "second_pass": t_pass2},
            "_statistics_kernels_ms": {"features": kt[0], "syrk": kt[1], "syrk_diag": kt[2]}, "_neg_elbo": float(res[0]),
            "_flops": {"per_row_statistics": fl_stats, "per_row_second_pass": fl_pass2, "per_row": fl_row, "posterior": fl_post},
            "parity": {"neg_elbo_256_rows": perr[0], "gradient_256_rows": perr[1]},
            "roofline": {"bound": "mfma", "peak": peak, "frac": (fl_row * N) / (t_eval * 1e-3) / 1e12 / peak,
                         "statistics_frac": fl_stats * N / (t_stats * 1e-3) / 1e12 / peak,
                         "second_pass_frac": fl_pass2 * N / (t_pass2 * 1e-3) / 1e12 / peak,
                         "posterior_frac_f64": fl_post / (t_post * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                         "_frac_over_stage_sum": (fl_row * N) / ((t_stats + t_post + t_pass2) * 1e-3) / 1e12 / peak}}


def config_posterior(dev, _hip, args, F):
    """rr_posterior_dev alone: C = (diag(1/L) + G / var)^-1 by blocked Cholesky + inverse on the f64 MFMA GEMMs, m, diag C,
    log|iC| and sum(G o C), all in HBM; G = the Gram of 3 F random-feature rows (full rank, realistic spectrum)."""
    rs = np.random.RandomState(F)
    nrows = 3 * F if F < 16384 else F // 4  # (F = 16384: the host's P^T P of 3 F rows alone would take half a minute)
    A = rs.standard_normal((nrows, 16)) @ rs.standard_normal((16, F)) / 4.0
    P = np.concatenate((np.cos(A), np.sin(A)), axis=1)[:, :F] / np.sqrt(F / 2.0)
    del A
    G = P.T @ P
    b = P.T @ rs.standard_normal(nrows)
    del P
    iL, var = np.full(F, 1.0), 0.5
    acc = dev.upload_vector(np.concatenate((G.ravel(), b)))
    dC = dev.malloc(F * F * 8)
    pG, pb = _hip.ctypes.c_void_p(acc.ptr.value), _hip.ctypes.c_void_p(acc.ptr.value + F * F * 8)
    dev.posterior(F, pG, pb, iL, var, dC)
    ms, post = _median_ms(lambda: dev.posterior(F, pG, pb, iL, var, dC))
    m, dg, logdet, tr = post
    perr = None
    if not args.no_parity_check and F < 16384:
        orc = _oracle()
        mh, Ch, ldC = orc.slm_posterior_from_stats(G, b, var, np.full(F, 1.0))
        C = dev.download(dC, (F, F), np.float64)
        trh = float((G * Ch).sum())
        perr = {"m": parity("m vs oracle solve_posdef", float(np.abs(m - mh).max() / np.abs(mh).max()), 1e-9),
                "C": parity("C vs oracle solve_posdef", float(np.abs(C - Ch).max() / np.abs(Ch).max()), 1e-9),
                "logdet_abs": parity("log|iC| vs oracle", float(abs(logdet + ldC)), 1e-8),
                "trace": parity("sum(G o C) vs oracle", float(abs(tr - trh) / abs(trh)), 1e-9)}
        del C, Ch
    elif not args.no_parity_check:
        # F = 16384: the oracle's full inverse takes the better part of a minute on the host (tests/test_gpu_posterior.py does
        # that comparison); here the size-independent properties of the SAME quantities, in host float64: iC C = I on 64
        # random columns, iC m = b / var, log|iC| from the host's Cholesky factor alone, sum(G o C) from the downloaded C
        import scipy.linalg as sla
        iC = G / var
        iC[np.diag_indices(F)] += iL
        C = dev.download(dC, (F, F), np.float64)
        cols = rs.choice(F, 64, replace=False)
        R = iC @ C[:, cols]
        R[cols, np.arange(64)] -= 1.0
        ldh = 2.0 * float(np.log(sla.cholesky(iC, lower=False, overwrite_a=False, check_finite=False).diagonal()).sum())
        trh = float((G * C).sum())
        perr = {"iC_C_minus_I_64_cols": parity("max|iC C - I| on 64 columns", float(np.abs(R).max()), 1e-9),
                "m": parity("|iC m - b / var| / |b / var|", float(np.abs(iC @ m - b / var).max() / np.abs(b / var).max()), 1e-9),
                "logdet_abs": parity("log|iC| vs the host's Cholesky", float(abs(logdet - ldh) / abs(ldh)), 1e-10),
                "trace": parity("sum(G o C) vs host sum over the downloaded C", float(abs(tr - trh) / abs(trh)), 1e-9),
                "C_symmetric": bool(np.array_equal(C, C.T))}
        del C, iC, R
    acc.free()
    dC.free()
    fl = posterior_flops(F)
    return {"workload": "rr_posterior_dev F=%d f64: Cholesky + inverse + m, diag C, log|iC|, sum(G o C) in HBM" % F,
            "ms": ms, "dtype": "f64", "parity": perr,
            "roofline": {"bound": "mfma", "peak": PEAK_F64_MFMA_TFLOPS, "frac": fl / (ms * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                         "frac_4F3_3_count": 4.0 / 3.0 * fl / (ms * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                         "_flops": fl, "_what": "F^3 = potrf F^3/3 + triangular inverse F^3/3 + triangular Y^T Y F^3/3 -- the "
                                                "algorithm executed (rr_posdef.hip:5-10) -- over the wall-clock of the whole call; "
                                                "frac_4F3_3_count is rounds 2-5's count (F^3/3 + a full F^3 inverse)"}}


def config_predict(dev, _hip, args, N=300_000):
    """StandardLinearModel.predict_moments (slm.py:219-244) for N query rows at F = 4096: host X in, (Ey, Vy) out."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    d, n = 32, 2048
    F = 2 * n
    wvec = np.random.RandomState(1).randn(d).astype(np.float32)
    X, y = gen_chunk(92, N, d, wvec)
    basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=42, lenscale=Parameter(1.0, Positive()))
    slm = StandardLinearModel(basis, var=Parameter(0.5, Positive()), nstarts=0, maxiter=1).fit(X[:50000], y[:50000])
    slm.predict_moments(X[:4096])
    ms, (Ey, Vy) = _median_ms(lambda: slm.predict_moments(X))
    ms_mean, _ = _median_ms(lambda: slm.predict(X))
    perr = None
    if not args.no_parity_check:
        orc = _oracle()
        Phi = orc.rff_transform(X[:512].astype(np.float64), basis.W, slm.hypers_)
        Er, Vr = orc.slm_predict_moments(Phi, slm.weights_, slm.covariance_, slm.var_)
        perr = {"Ey": parity("Ey of 512 rows vs oracle", float(np.abs(Ey[:512] - Er).max() / np.abs(Er).max()), 1e-3),
                "Vy": parity("Vy of 512 rows vs oracle", float(np.abs(Vy[:512] - Vr).max() / np.abs(Vr).max()), 1e-3)}
    slm._drop_serving()
    fl = 2.0 * d * n + F * F + 2.0 * F  # features, phi^T B with the triangular factor (half of 2 F^2), phi . m
    return {"workload": "predict_moments, RandomRBF F=4096 D=32, N=%d HOST rows in, (Ey, Vy) out (PCIe inside)" % N,
            "rows": N, "dtype": "f32", "ms": ms, "ms_predict_mean_only": ms_mean, "value": N / (ms * 1e-3), "unit": "rows/s",
            "parity": perr,
            "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "frac": fl * N / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "_flops_per_row": fl}}


def config_predict_c3(dev, _hip, args, N=300_000):
    """The same call for config 3's concatenation (RandomMatern52 n=4096 + LinearBasis, F_tot = 8257, D = 64):
    BasisCat.predict_moments assembles Phi child by child in 65 536-row chunks of the host query."""
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    from revrand_amd.utils import atleast_list
    d, n = 64, 4096
    X, y = _c3_data(N, d, 7)
    basis = _c3_basis(d, n)
    slm = StandardLinearModel(basis, var=Parameter(0.5, Positive()), nstarts=0, maxiter=1).fit(X[:50000], y[:50000])
    F = slm.weights_.shape[0]
    slm.predict_moments(X[:4096])
    ms, (Ey, Vy) = _median_ms(lambda: slm.predict_moments(X))
    ms_mean, _ = _median_ms(lambda: slm.predict(X))
    perr = None
    if not args.no_parity_check:
        orc = _oracle()
        Phi, _, _ = _oracle_features(basis, X[:256].astype(np.float64), atleast_list(slm.hypers_))
        Er, Vr = orc.slm_predict_moments(Phi, slm.weights_, slm.covariance_, slm.var_)
        perr = {"Ey": parity("Ey of 256 rows vs oracle", float(np.abs(Ey[:256] - Er).max() / np.abs(Er).max()), 1e-3),
                "Vy": parity("Vy of 256 rows vs oracle", float(np.abs(Vy[:256] - Vr).max() / np.abs(Vr).max()), 1e-3)}
    slm._drop_serving()
    fl = 2.0 * d * n + float(F) * F + 2.0 * F
    return {"workload": "predict_moments, RandomMatern52 n=4096 + LinearBasis F_tot=%d D=64, N=%d HOST rows in (PCIe inside)" % (F, N),
            "rows": N, "dtype": "f32", "ms": ms, "ms_predict_mean_only": ms_mean, "value": N / (ms * 1e-3), "unit": "rows/s",
            "parity": perr,
            "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "frac": fl * N / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "_flops_per_row": fl}}


# ----------------------------------------------------------------------------------------------------
# N > 1: BASELINE config 3 and the C2-shape `_elbo` with the rows sharded over the ranks (slm.py:142-199 over all shards)
# ----------------------------------------------------------------------------------------------------

def dist_elbo(dev, _hip, comm, args, make_basis, gen, N, d, n_rff, hyp, reg, var=0.5, reps=2, parity_fn=None, flops=None):
    """One distributed `_elbo` evaluation, every stage on the ranks' clocks (barrier before, MAX over ranks after):
      statistics   this rank's features + Gram kernels (no exchange)
      exchange     pack the upper triangle, ONE ncclAllReduce of [tri G | b | y^T y | N], unpack + mirror  (HIP events)
      posterior    replicated rr_posterior_dev of the summed statistics
      second_pass  this rank's rows again (Err, U = Phi C in registers, X^T A) + the all-reduce of [sqErr | dhyp]
      ms           StandardLinearModel(distributed=True)._elbo as the optimiser calls it
    `gen(rows, stream)` makes chunk `stream` of the data set (250 000-row chunks, so the data do not depend on the world
    size); every rank takes its contiguous shard."""
    from revrand_amd import parallel
    from revrand_amd.slm import StandardLinearModel
    rank, world = comm.rank, comm.world
    t_last = [time.perf_counter()]

    def tick(what):  # RR_BENCH_TRACE=1: where a configuration's wall-clock goes (stderr)
        if os.environ.get("RR_BENCH_TRACE") == "1" and rank == 0:
            now = time.perf_counter()
            sys.stderr.write("bench.py: trace: %-28s %8.2f s\n" % (what, now - t_last[0]))
            t_last[0] = now
    CH = 250_000
    r0, r1 = parallel.shard_bounds(N, rank, world)
    Xs, ys = [], []
    for c in range(r0 // CH, (r1 + CH - 1) // CH if r1 > r0 else 0):
        Xc, yc = gen(min(CH, N - c * CH), c)
        lo, hi = max(r0, c * CH) - c * CH, min(r1, (c + 1) * CH) - c * CH
        Xs.append(Xc[lo:hi])
        ys.append(yc[lo:hi])
    X, y = np.ascontiguousarray(np.concatenate(Xs)), np.ascontiguousarray(np.concatenate(ys))
    del Xs, ys
    rows = r1 - r0
    tick("data")
    basis = make_basis()
    slm = StandardLinearModel(basis, distributed=True)
    slm.obj_ = -np.inf
    slm._defer_cov = True
    st = slm._state = slm._make_state(X, y)
    F = st.F
    regs = reg if isinstance(reg, list) else [reg]
    L, _ = basis.regularizer_diagonal(X, *regs)
    iL = 1.0 / L

    def rank_max(ms):
        return float(comm.allreduce_host(np.array([ms]), op="max")[0])

    def stage(fn, reps_=reps):
        ts, out = [], None
        for _ in range(reps_):
            dev.sync()
            comm.barrier()
            t0 = time.perf_counter()
            out = fn()
            dev.sync()
            ts.append(rank_max(1e3 * (time.perf_counter() - t0)))
        return float(np.median(ts)), out

    f0, _ = slm._elbo(X, y, var, reg, hyp)  # warm: scratch, posterior work space, RCCL channels
    tick("state + warm _elbo")
    t_stats, _ = stage(lambda: st.gram_device(hyp))
    tick("statistics")
    # the exchange alone, on the context's stream between HIP events (the statistics are re-made afterwards)
    pG, pb, pt = st._stat_ptrs()
    xs = []
    for _ in range(reps):
        dev.sync()
        comm.barrier()
        dev.timer_start()
        comm.reduce_stats_device(F, pG, pb, pt, rows, wait=False)
        xs.append(rank_max(dev.timer_stop()))
    t_x = float(np.median(xs))
    st.gram_device(hyp, comm.reduce_stats_device)
    N_tot = st.N_total
    # size-independent properties of the summed statistics: every rank holds all N rows' worth; trace of the Fourier
    # block == N (cos^2 + sin^2 = 1 per frequency)
    tick("exchange")
    G, _, _ = st.stats_host()
    tr = abs(float(np.trace(G[:2 * n_rff, :2 * n_rff])) - N) / N
    sym = bool(np.array_equal(G, G.T))
    del G
    tick("stats_host + trace")
    t_post, post = stage(lambda: st.posterior(iL, var))
    assert post is not None, "posterior not positive definite"
    m = post[0]

    tick("posterior")
    def pass2():
        sq, dh = st.second_pass(hyp, m, st.dC, var)
        parts = dh if isinstance(dh, list) else [dh]
        return comm.allreduce_host(np.concatenate([[sq]] + [np.atleast_1d(p) for p in parts]))
    t_p2, _ = stage(pass2)
    tick("second pass")
    t_eval, res = stage(lambda: slm._elbo(X, y, var, reg, hyp))
    # every rank walked to the same numbers (rank 0's are broadcast unless the reductions are deterministic)
    flat = np.concatenate([[res[0]], np.atleast_1d(res[1][0]), np.ravel(np.atleast_1d(res[1][1])),
                           np.ravel(np.concatenate([np.atleast_1d(h) for h in (res[1][2] if isinstance(res[1][2], list) else [res[1][2]])]))])
    same = bool(np.array_equal(comm.allreduce_host(flat, op="max"), comm.allreduce_host(flat, op="min")))
    tick("_elbo + identity")
    st.release()
    slm._state = None
    perr = (None, None)
    if rank == 0 and not args.no_parity_check:  # the same evaluation in ONE process on 256 rows against the oracle
        prev = parallel.get_comm()
        parallel.set_comm(parallel.SingleComm())
        try:
            perr = (parity_fn or _elbo_parity)(make_basis, X, y, var, reg, hyp)
        finally:
            parallel.set_comm(prev)
    tick("oracle parity")
    d_ = d
    fl_stats = 2.0 * d_ * n_rff + F * (F + 1.0) + 2.0 * F
    fl_p2 = 2.0 * F * F + 4.0 * d_ * n_rff
    if flops is not None:   # (a basis whose features are not a d x n product: FastFood's chain)
        fl_stats, fl_p2 = flops(F)
    peak = world * PEAK_F32_MFMA_TFLOPS
    out = {"rows": N, "rows_per_gpu": rows, "F": F, "dtype": "f32", "ms": t_eval, "value": N / (t_eval * 1e-3),
           "unit": "rows/s per _elbo",
           "stage_ms": {"statistics": t_stats, "exchange": t_x, "posterior": t_post, "second_pass": t_p2},
           "exchange_bytes": 8 * parallel.stats_count(F),
           "exchange_GBps_busbw": 2.0 * (world - 1) / world * 8 * parallel.stats_count(F) / (t_x * 1e-3) / 1e9 if t_x > 0 else None,
           "parity": {"trace_fourier_block": tr, "G_symmetric": sym, "N_total": N_tot, "ranks_identical": same,
                      "neg_elbo_256_rows": perr[0], "gradient_256_rows": perr[1]},
           "roofline": {"bound": "mfma", "peak": peak, "frac": (fl_stats + fl_p2) * N / (t_eval * 1e-3) / 1e12 / peak,
                        "statistics_frac": fl_stats * N / (t_stats * 1e-3) / 1e12 / peak,
                        "second_pass_frac": fl_p2 * N / (t_p2 * 1e-3) / 1e12 / peak,
                        "posterior_frac_f64_one_gpu": posterior_flops(F) / (t_post * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS},
           # one GPU = every shard's row passes back to back + one posterior; N GPUs = the largest shard + the exchanges +
           # the (replicated, serial) posterior
           "speedup_model": {"one_gpu_ms": world * (t_stats + t_p2) + t_post, "n_gpu_ms": t_stats + t_x + t_post + t_p2,
                             "speedup": (world * (t_stats + t_p2) + t_post) / (t_stats + t_x + t_post + t_p2)},
           "_neg_elbo": float(res[0])}
    if rank == 0:
        parity("trace of the Fourier block / N - 1", tr, 1e-5)
        assert sym and N_tot == N and same, (sym, N_tot, same)
        if perr[0] is not None:
            parity("-ELBO of 256 rows vs oracle", perr[0], 1e-4)
            parity("gradient of 256 rows vs oracle (normwise)", perr[1], 2e-3)
    return out


def _c4_data(rows, c):
    """Chunk c of config 4's synthetic data set (D = 128), the same on every world size."""
    rng = np.random.default_rng([20260928, 44, c])
    X = rng.standard_normal((rows, 128), dtype=np.float32)
    w = np.random.default_rng([20260928, 45]).standard_normal(128, dtype=np.float32)
    y = (np.sin(X @ w / np.sqrt(128.0)) + 0.1 * rng.standard_normal(rows, dtype=np.float32)).astype(np.float32)
    return X, y


def dist_configs(dev, _hip, comm, args, emit=None):
    """BASELINE config 3 (RandomMatern52 + LinearBasis, F_tot = 8257, D = 64, N = 10M, "8 GPUs N-sharded with RCCL Gram
    all-reduce") and the headline shape's `_elbo` (RandomRBF F = 4096, D = 32, N = 10M), rows sharded over the ranks."""
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    res = {}
    want = args.configs.lower().split(",")

    def rbf_gen(rows, c):
        return gen_chunk(c, rows, 32, np.random.RandomState(1).randn(32).astype(np.float32))

    jobs = (("C3_matern52_linear_dist", lambda: dist_elbo(
                dev, _hip, comm, args, lambda: _c3_basis(64, 4096), lambda rows, c: _c3_data(rows, 64, c),
                args.dist_rows, 64, 4096, [np.ones(64)], [1.0, 1.0])),
            ("elbo_rbf_f4096_dist", lambda: dist_elbo(
                dev, _hip, comm, args, lambda: bs.RandomRBF(nbases=2048, Xdim=32, random_state=42,
                                                            lenscale=Parameter(np.ones(32), Positive())),
                rbf_gen, args.dist_rows, 32, 2048, np.linspace(0.8, 1.3, 32), 1.0)),
            # BASELINE config 4 with the Gram (C4 is row-sharded too): FastFoodRBF F = 16384, D = 128, the chain
            # kernel into every rank's feature matrix, the 2 GiB-per-rank statistics summed as ONE 1.07 GB message
            ("C4elbo_fastfood_f16384_dist", lambda: dist_elbo(
                dev, _hip, comm, args, lambda: bs.FastFoodRBF(nbases=8192, Xdim=128, random_state=1,
                                                              lenscale=Parameter(np.ones(128), Positive())),
                _c4_data, args.dist_rows_c4, 128, 8192, np.linspace(0.8, 1.3, 128), 1.0, parity_fn=_ff_elbo_parity,
                flops=lambda F: (F * (F + 1.0) + 2.0 * F + 64 * (2.0 * 128 * 7 + 3.0 * 128), 2.0 * F * F + 2.0 * 128 * F))),
            ("C5_glm_svi_step_dist", lambda: dist_glm(dev, _hip, comm, args)))
    for name, fn in jobs:
        if args.configs != "all" and name.split("_")[0].lower() not in want and name.lower() not in want:
            continue
        t0 = time.perf_counter()
        sys.stderr.write("bench.py: [rank %d] config %s ...\n" % (comm.rank, name))
        sys.stderr.flush()
        # A rank that fails inside a stage leaves its peers waiting in the next collective: every rank arms a watchdog
        # that, after --config-timeout, writes the line with what is finished (rank 0) and ends the process.
        dog = _watchdog(args, name, res, emit)
        try:
            out, err, fatal = fn(), None, 0.0
        except ParityError as e:  # raised after the configuration's last collective: the ranks are still in step
            out, err, fatal = None, str(e), 0.0
        except Exception as e:
            out, err, fatal = None, "%s: %s" % (type(e).__name__, e), 1.0
            sys.stderr.write("bench.py: [rank %d] config %s failed: %r\n" % (comm.rank, name, e))
        try:
            fatal = float(comm.allreduce_host(np.array([fatal]), op="max")[0])
        finally:
            dog.cancel()
        res[name] = out if err is None else {"error": err}
        if out is not None:
            out["_bench_seconds"] = time.perf_counter() - t0
        sys.stderr.write("bench.py: [rank %d] config %s done in %.1f s\n" % (comm.rank, name, time.perf_counter() - t0))
        if fatal:
            break
    return res


def dist_glm(dev, _hip, comm, args):
    """Config 5's SVI step with one process per GPU (`GeneralizedLinearModel(distributed=True)`): the 2 M rows sharded over the
    ranks, every rank cuts 65 536 / world rows of ITS shard per step (the job's minibatch stays 65 536 rows), the loop resident
    on every rank (rr_glm_sgd_dist_step: dT and [Edm | EdC | sums | llconst | rows] all-reduced over the ranks in HBM, the
    update replicated).  The interval between the moments rank 0 queues its steps over steps 8.. of a 56-step fit, device
    sampler; the ranks' parameters after the fit must be the same bits."""
    import logging
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd import parallel
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    logging.getLogger("revrand_amd").setLevel(logging.ERROR)
    N, d, n, K, L, M = _glm_rows(args), 32, 1024, 10, 50, 65536
    world, rank = comm.world, comm.rank
    a, b = parallel.shard_bounds(N, rank, world)
    rng = np.random.default_rng([20260928, 5, rank])
    X = rng.standard_normal((b - a, d), dtype=np.float32)
    y = rng.poisson(np.exp(0.3 * X[:, 0].astype(np.float64))).astype(np.float64)

    def fit(iters):
        g = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                                   K=K, nsamples=L, batch_size=M // world, maxiter=iters, nstarts=0, random_state=2, sampler="device",
                                   distributed=True)
        np.random.seed(20260930)
        g.fit(X, y)
        flat = np.concatenate((g.weights_.ravel(), g.covariance_.ravel(), np.atleast_1d(g.basis_hypers_)))
        return flat, g.__dict__.get("_resident_clock")
    fit(8)
    flat, ck = fit(56)
    resident = ck is not None and len(ck) > 12
    ms = float(np.median(1e3 * np.diff(ck[8:-1]))) if resident else float("nan")
    ms = float(comm.allreduce_host(np.array([ms]), op="max")[0])
    # every rank holds the same parameters: max and min over the ranks of a few checksums coincide
    chk = np.array([flat.sum(), np.abs(flat).sum(), float(flat[:: max(1, flat.size // 997)] @ np.arange(len(flat[:: max(1, flat.size // 997)])))])
    hi, lo = comm.allreduce_host(chk, op="max"), comm.allreduce_host(chk, op="min")
    same = bool(np.array_equal(hi, lo)) and bool(np.all(np.isfinite(flat)))
    gemm_flops = 3 * 2.0 * K * L * M * 2 * n
    out = {"workload": "config 5's SVI step, fit(distributed=True): N=%d over %d ranks, job minibatch 65536, device sampler" % (N, world),
           "ms": ms, "value": M / (ms * 1e-3), "unit": "minibatch-rows/s", "dtype": "f32", "resident_loop": bool(resident),
           "parity": {"ranks_identical": same},
           "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS * world,
                        "frac": gemm_flops / (ms * 1e-3) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)}}
    if not same:
        raise ParityError("C5 distributed fit: the ranks' parameters differ")
    return out


def _watchdog(args, name, res, emit):
    """A configuration that hangs (a HIP call, a collective or a BLAS call that never returns cannot be interrupted from
    Python) says where and does not take the headline with it: after --config-timeout seconds every thread's stack goes to
    stderr, the bench line is written with the configurations finished so far -- this one marked as timed out -- and
    the process ends."""
    import threading

    def on_timeout():
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        res[name] = {"error": "timed out after %.0f s (stacks on stderr); the remaining configurations were not run"
                              % args.config_timeout}
        sys.stderr.write("bench.py: config %s timed out\n" % name)
        sys.stderr.flush()
        if emit is not None:
            emit(res)
        os._exit(0 if emit is not None else 1)
    dog = threading.Timer(args.config_timeout, on_timeout)
    dog.daemon = True
    dog.start()
    return dog


def extra_configs(dev, _hip, args, emit=None):
    """`emit(configs)`: writes the bench line with the configurations finished so far -- called by the watchdog when
    one of them hangs, so that the headline measurement is never lost to a side configuration."""
    res = {}
    for name, fn in (("C1_elbo_latency", config_c1), ("C2_rbf_f4096_n1m", config_c2), ("headline_shape_f64", config_f64),
                     ("C2laplace_f64phase_n1m", config_laplace), ("C2_elbo_eval", config_elbo),
                     ("C2f64_elbo_eval_n200k", lambda d_, h_, a_: config_elbo(d_, h_, a_, dtype="f64", N=200_000)),
                     ("posterior_F4096", lambda d_, h_, a_: config_posterior(d_, h_, a_, 4096)),
                     ("posterior_F8257", lambda d_, h_, a_: config_posterior(d_, h_, a_, 8257)),
                     ("posterior_F16384", lambda d_, h_, a_: config_posterior(d_, h_, a_, 16384)),
                     ("predict_moments_n300k", config_predict), ("predictc3_moments_n300k", config_predict_c3),
                     ("C3_matern52_linear_concat_one_gpu_share", config_c3), ("C4_fastfood_f16384", config_c4),
                     ("C4elbo_fastfood_f16384", config_ff_elbo), ("C4gm_fastfoodgm_f16384", config_c4gm), ("C5_glm_poisson_svi_step", config_c5),
                     ("C5small_glm_default_fit", config_glm_default)):
        want = args.configs.lower().split(",")
        if args.configs != "all" and name.split("_")[0].lower() not in want and name.lower() not in want:
            continue
        t0 = time.perf_counter()
        sys.stderr.write("bench.py: config %s ...\n" % name)
        sys.stderr.flush()
        dog = _watchdog(args, name, res, emit)
        try:
            if os.environ.get("RR_BENCH_TEST_HANG", "").lower() == name.lower():  # tests/test_gpu_comm.py: the watchdog's own test
                time.sleep(1e6)
            res[name] = fn(dev, _hip, args)
            res[name]["_bench_seconds"] = time.perf_counter() - t0
            # the line carries numbers (the driver keeps a 9 KB tail): a side configuration's prose -- what it ran, what its
            # cpu sample was -- goes to the unabridged record (--full-json); the configuration's NAME says which it is
            if "workload" in res[name]:
                res[name]["_workload"] = res[name].pop("workload")
            cb = res[name].get("cpu_baseline")
            if isinstance(cb, dict) and "sample" in cb:
                cb["_sample"] = cb.pop("sample")
        except Exception as e:  # a failing side configuration must not take the headline line with it -- but it is said
            res[name] = {"error": "%s: %s" % (type(e).__name__, e) if not isinstance(e, ParityError) else str(e)}
            sys.stderr.write("bench.py: config %s failed: %r\n" % (name, e))
        finally:
            dog.cancel()
        sys.stderr.write("bench.py: config %s done in %.1f s\n" % (name, time.perf_counter() - t0))
        sys.stderr.flush()
    return res


# ----------------------------------------------------------------------------------------------------
# where the GPUs sit, and what the link between them moves -- measured before anything is timed
# ----------------------------------------------------------------------------------------------------

XGMI_LINK_GBPS = 153.0  # MI355X_MICROARCH.md: per-link, per-direction xGMI bandwidth of the 8-GPU mesh (7 links per GPU)


def placement(dev, pin=True):
    """PCI bus id and host memory node of a context's GPU; pin: the calling process's threads go to that node's CPUs (a
    rank's staging copies and its launch thread then sit next to its GPU).  Never fatal."""
    info = {"device": dev.index}
    try:
        info["pci"] = dev.pci_bus_id
        info["numa"] = dev.numa_node
        cpus = dev.numa_cpus()
        if pin and cpus and hasattr(os, "sched_setaffinity") and os.environ.get("RR_BENCH_NO_PIN") != "1":
            allowed = os.sched_getaffinity(0) & cpus
            if allowed:
                os.sched_setaffinity(0, allowed)
                info["pinned_cpus"] = len(allowed)
                try:  # BLAS made its thread pool for every CPU of the box at import: as many as the pinned CPUs now.
                    # Only ever LOWERED: a launcher may have started us with OMP_NUM_THREADS=1 (torch.distributed.run does),
                    # and OpenBLAS asked for more threads than it was initialised with crashes in its next LAPACK call
                    from threadpoolctl import threadpool_info, threadpool_limits
                    cur = max([p_.get("num_threads", 1) for p_ in threadpool_info()] or [1])
                    if cur > len(allowed):
                        threadpool_limits(limits=len(allowed))
                except Exception:
                    pass
    except Exception as e:  # a container without /sys, a restricted cpuset, ...
        info["_error"] = "%s: %s" % (type(e).__name__, e)
    return info


def exchange_model_ms(nbytes, world):
    """DESIGN 5's model of one all-reduce of S bytes over the xGMI mesh: 2 (n-1)/n S per GPU at one link's rate (a ring; the
    direct peer algorithm spreads the same bytes over all n-1 links)."""
    return 2.0 * (world - 1) / world * nbytes / (XGMI_LINK_GBPS * 1e9) * 1e3 if world > 1 else 0.0


def preflight_exchange(reduce_fn, sync_fn, timer_dev, count, world, reps=3):
    """One warm-up and `reps` timed in-place all-reduces of `count` float64 (config 3's 273 MB message by default) BEFORE
    anything else is timed: the first collective over a transport this repository has never been measured on must not be
    inside a timed region, and its bus bandwidth says at once whether the ranks talk over xGMI (P2P), PCIe or sockets."""
    nbytes = 8 * count
    reduce_fn()
    sync_fn()
    ms = []
    for _ in range(reps):
        sync_fn()
        timer_dev.timer_start()
        reduce_fn()
        ms.append(timer_dev.timer_stop())
        sync_fn()
    best = float(min(ms))
    busbw = 2.0 * (world - 1) / world * nbytes / (best * 1e-3) / 1e9 if world > 1 and best > 0 else None
    model = exchange_model_ms(nbytes, world)
    return {"message_bytes": nbytes, "ms": best, "ms_all": [float(m) for m in ms], "busbw_GBps": busbw,
            "model_ms_at_one_xgmi_link": model, "busbw_frac_of_one_link": (busbw / XGMI_LINK_GBPS) if busbw else None}


# ----------------------------------------------------------------------------------------------------
# --single-process: the same step, N GPUs behind ONE process (StandardLinearModel(devices=...)'s machinery)
# ----------------------------------------------------------------------------------------------------

def single_process(args, json_out):
    """`python bench.py --gpus N --single-process`: the headline step with the row shards on the members of an in-process
    device group (revrand_amd/multigpu.py: one context and one host thread per GPU, ONE collective per step through
    rr_comm_group_reduce_stats_dev) -- what `StandardLinearModel(basis, devices=N).fit` runs -- then one `_elbo` of that
    estimator.  Same line shape as the one-process-per-GPU run.  With fewer GPUs than members (the 1-GPU test box) members
    share devices: plumbing only, flagged "oversubscribed"."""
    from revrand_amd import _hip, multigpu, parallel
    ndev = multigpu.visible_devices()
    world = args.gpus
    devices = list(range(world)) if ndev >= world else [r % ndev for r in range(world)]
    group = multigpu.get_group(devices)
    d, n = args.dim, args.nbases
    F = 2 * n
    W = np.random.RandomState(42).randn(d, n)
    wvec = np.random.RandomState(1).randn(d).astype(np.float32)
    if args.engine:
        group.set_gram_engine(args.engine)
    place = [{"device": m.index, "pci": m.pci_bus_id, "numa": m.numa_node} for m in group.members]
    peers_ok = all(a.can_access_peer(b.index) for a in group.members for b in group.members)
    CH = 250_000
    bounds = [parallel.shard_bounds(args.rows, i, world) for i in range(world)]
    nacc = F * F + F + 1

    class Shard(object):
        pass

    def build(i):
        dev = _hip.get_device()  # member i's context (this is its thread)
        sh = Shard()
        sh.dev = dev
        sh.basis = _hip.RffHandle(W, compute="f32")
        row0, row1 = bounds[i]
        sh.rows = row1 - row0
        sh.dX = dev.empty_matrix(sh.rows, d, np.float32, ld_dev=sh.basis.padded_dim)
        sh.dy = dev.malloc(max(sh.rows, 1) * 4)
        sh.dy.dtype = np.dtype(np.float32)
        r0 = 0
        for c in range(row0 // CH, (row1 + CH - 1) // CH if sh.rows else 0):
            c0 = c * CH
            Xc, yc = gen_chunk(c, min(CH, args.rows - c0), d, wvec)
            lo, hi = max(row0, c0) - c0, min(row1, c0 + CH) - c0
            Xc, yc = np.ascontiguousarray(Xc[lo:hi]), np.ascontiguousarray(yc[lo:hi])
            dev.upload_rows(sh.dX, r0, Xc)
            _hip._check(dev.lib, dev.lib.rr_memcpy_h2d(dev.ctx, _hip.ctypes.c_void_p(sh.dy.ptr.value + r0 * 4),
                                                       yc.ctypes.data_as(_hip.ctypes.c_void_p), yc.nbytes))
            r0 += hi - lo
        sh.acc = dev.zeros(nacc * 8)
        base = sh.acc.ptr.value
        sh.ptrs = tuple(_hip.ctypes.c_void_p(base + o * 8) for o in (0, F * F, F * F + F))
        return sh
    shards = group.map(build)
    m0 = group.members[0]

    # ---- preflight: config 3's message through the group's collective, before anything is timed ----
    cnt3 = parallel.stats_count(8257)
    pre_bufs = group.map(lambda i: _hip.get_device().zeros(cnt3 * 8))
    pre = preflight_exchange(lambda: group.allreduce_device(pre_bufs, cnt3), group.sync, m0, cnt3, world)
    for b in pre_bufs:
        b.free()
    kernel_ms, exch_ms = [], []

    def launch(i):
        sh = shards[i]
        _hip._check(sh.dev.lib, sh.dev.lib.rr_memset(sh.dev.ctx, sh.ptrs[0], 0, nacc * 8))
        if sh.rows:
            sh.basis.gram_dev(sh.dX, sh.dy, 1.0, *sh.ptrs)

    def step(timed):
        group.map(launch)
        if timed:
            m0.timer_start()
        group.reduce_stats(F, [sh.ptrs for sh in shards], [sh.rows for sh in shards], wait=False)
        if timed:
            exch_ms.append(m0.timer_stop())
        group.sync()
        if timed:
            kernel_ms.append([sh.basis.gram_timings() for sh in shards])

    for _ in range(args.warmup):
        step(False)
    group.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    group.sync()
    elapsed = time.perf_counter() - t0
    # every member on its own GPU: the same message through the library's OWN transport too (reduce-scatter + all-gather
    # kernels loading the peers' HBM over the xGMI mesh) -- a second group of contexts on the same GPUs, after the timed
    # steps, on a thread that is given 90 s: a transport that has never run between two GPUs must not take the line with it
    pre_peer, peer_hung = None, False
    if (group.transport == "rccl" or os.environ.get("RR_BENCH_FORCE_PEER_PREFLIGHT") == "1") \
            and os.environ.get("RR_BENCH_NO_PEER_PREFLIGHT") != "1":
        import threading
        box = {}

        def peer_preflight():
            try:
                pg = multigpu.get_group(devices, transport="peer")
                pb = pg.map(lambda i: _hip.get_device().zeros(cnt3 * 8))
                res = preflight_exchange(lambda: pg.allreduce_device(pb, cnt3), pg.sync, pg.members[0], cnt3, world)
                ones = pg.map(lambda i: _hip.get_device().upload_vector(np.full(4099, float(i + 1))))
                pg.allreduce_device(ones, 4099)
                got = [m.download(o, (4099,), np.float64) for m, o in zip(pg.members, ones)]
                res["sum_exact"] = bool(all(np.array_equal(g_, np.full(4099, world * (world + 1) / 2.0)) for g_ in got))
                for b_ in pb + ones:
                    b_.free()
                box["res"] = res
            except Exception as e:  # noqa: BLE001 -- the line says so
                box["res"] = {"error": "%s: %s" % (type(e).__name__, e)}
        th = threading.Thread(target=peer_preflight, name="rr-peer-preflight", daemon=True)
        th.start()
        th.join(90.0)
        peer_hung = th.is_alive()
        pre_peer = {"error": "no result after 90 s"} if peer_hung else box.get("res")
    Gs = [sh.dev.download(sh.acc, (F, F), np.float64) for sh in shards]
    G = Gs[0]
    trace_err = abs(float(np.trace(G)) - args.rows) / args.rows
    members_identical = all(np.array_equal(G, Gi) for Gi in Gs[1:])
    assert np.array_equal(G, G.T)
    del Gs, G
    per_member = [float(np.mean([k[i][0] + k[i][1] + k[i][2] for k in kernel_ms])) for i in range(world)]
    syrk0 = float(np.mean([k[0][1] for k in kernel_ms]))
    launches = kernel_ms[0][0][3]
    widths = [min(256, F - 256 * i) for i in range((F + 255) // 256)]
    off_flops = 2.0 * sum(widths[i] * widths[j] for i in range(len(widths)) for j in range(i + 1, len(widths)))
    achieved = off_flops * shards[0].rows / (syrk0 * 1e-3) / 1e12
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    cnt = parallel.stats_count(F)
    xm = float(np.mean(exch_ms))
    rt = runtime_info(_hip, parallel)
    out = {
        "metric": "feature-rows/sec (Phi + PhiT Phi + PhiT y) at N=%s D=%d F=%d" % (
            "10M" if args.rows == 10_000_000 else args.rows, d, F),
        "value": args.rows / (elapsed / max(args.steps, 1)), "unit": "feature-rows/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "RandomRBF nbases=%d (F=%d), D=%d, N=%d f32, features + MFMA Gram, rows sharded over %d GPU(s) "
                               "of ONE process (devices=%s)" % (n, F, d, args.rows, world, devices),
                   "single_process": True, "rows_per_gpu": shards[0].rows, "device": m0.name.strip(), "trace_rel_err": trace_err,
                   "members_bit_identical": bool(members_identical), "gram_engine": m0.gram_engine,
                   "runtime": {"hip_runtime": os.path.basename(rt["hip_runtime"] or ""),
                               "rccl": rt["rccl"].get("version", rt["rccl"].get("error"))}},
        "roofline": {"bound": "mfma", "kernel": "rr_syrk_f32_kernel", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                     "kernel_ms_per_step": syrk0, "launches_per_step": launches, "avg_launch_ms": syrk0 / max(launches, 1),
                     "flops_per_row": off_flops, "rows_per_step": shards[0].rows,
                     "whole_path_frac": flops_per_row(d, n) * args.rows / (elapsed / max(args.steps, 1)) / world / 1e12
                     / PEAK_F32_MFMA_TFLOPS},
        "exchange": {"transport": group.transport, "message_float64": cnt, "message_bytes": 8 * cnt,
                     "ms_per_step_pack_allreduce_unpack": xm,
                     "busbw_GBps": 2.0 * (world - 1) / world * 8 * cnt / (xm * 1e-3) / 1e9 if world > 1 and xm > 0 else None,
                     "model_ms_at_one_xgmi_link": exchange_model_ms(8 * cnt, world),
                     "visible_gpus": ndev, "distinct_gpus": len(set(devices)), "oversubscribed": len(set(devices)) < world,
                     "peer_access": bool(peers_ok), "preflight": pre, "preflight_peer_transport": pre_peer, "placement": place},
        "per_rank": {"kernel_ms_per_step_max_min_over_members": [max(per_member), min(per_member)],
                     "kernel_ms_per_step_sum_over_members": float(sum(per_member)),
                     "expected_speedup_model": {"ms_per_step": max(per_member) + xm,
                                                "speedup_vs_one_gpu": float(sum(per_member)) / (max(per_member) + xm),
                                                "measured_ms_per_step": ms_per_step}},
    }
    ok = trace_err < 1e-6 and members_identical

    def emit(configs=None):
        if configs is not None:
            out["configs"] = configs
        json_out.write(json.dumps(lean(out), separators=(",", ":")) + "\n")
        json_out.flush()
        path = args.full_json
        if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            path = os.path.join(ROOT, "gpurun_out", "bench_full_sp%d.json" % world)
        if path:
            try:
                with open(path, "w") as f:
                    json.dump(full(out), f, indent=1)
            except OSError as e:
                sys.stderr.write("bench.py: could not write %s: %s\n" % (path, e))

    def free(i):
        sh = shards[i]
        for b in (sh.dX, sh.dy, sh.acc):
            b.free()
        sh.basis = None
    group.map(free)
    configs = {}
    if args.configs != "none" and ok:
        dog = _watchdog(args, "elbo_rbf_f4096_single_process", configs, emit)
        try:
            configs["elbo_rbf_f4096_single_process"] = single_process_elbo(args, devices, group)
        except Exception as e:  # noqa: BLE001 -- the line still goes out, with the failure named
            configs["elbo_rbf_f4096_single_process"] = {"error": "%s: %s" % (type(e).__name__, e)}
        dog.cancel()
        dog = _watchdog(args, "C5_glm_svi_step_single_process", configs, emit)
        try:
            configs["C5_glm_svi_step_single_process"] = single_process_glm(args, devices, group)
        except Exception as e:  # noqa: BLE001
            configs["C5_glm_svi_step_single_process"] = {"error": "%s: %s" % (type(e).__name__, e)}
        dog.cancel()
    emit(configs or None)
    assert ok, (trace_err, members_identical)
    if peer_hung:
        os._exit(0)  # (a device call that never returns would hold the interpreter's exit)


def single_process_elbo(args, devices, group):
    """`StandardLinearModel(RandomRBF F = 4096, devices=...)._elbo` as the optimiser calls it, --dist-rows rows sharded over
    the members: per-stage wall-clock (statistics incl. the in-process exchange, replicated posterior, second pass), and the
    same evaluation on 256 rows against the oracle."""
    import revrand_amd.basis_functions as bs
    from revrand_amd import _hip
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel
    d, n = 32, 2048
    F = 2 * n
    N = args.dist_rows
    wvec = np.random.RandomState(1).randn(d).astype(np.float32)
    CH = 250_000
    parts = [gen_chunk(c, min(CH, N - c * CH), d, wvec) for c in range((N + CH - 1) // CH)]
    X = np.concatenate([p[0] for p in parts])
    y = np.concatenate([p[1] for p in parts])
    del parts

    def make_basis():
        return bs.RandomRBF(nbases=n, Xdim=d, random_state=42, lenscale=Parameter(np.ones(d), Positive()))
    slm = StandardLinearModel(make_basis(), devices=devices)
    slm.obj_ = -np.inf
    slm._defer_cov = True
    st = slm._state = slm._make_state(X, y)
    assert st is not None and hasattr(st, "states")
    ls, var, reg = np.linspace(0.8, 1.3, d), 0.5, 1.0
    iL = np.full(F, 1.0 / reg)
    slm._elbo(X, y, var, reg, ls)  # warm: scratch, posterior work space on every member
    t_stats, _ = _median_ms(lambda: st.gram_device(ls), reps=2)
    t_post, post = _median_ms(lambda: st.posterior(iL, var), reps=2)
    t_pass2, _ = _median_ms(lambda: st.second_pass(ls, post[0], st.dC, var), reps=2)
    t_eval, res = _median_ms(lambda: slm._elbo(X, y, var, reg, ls * 1.0), reps=2)
    G0 = st.states[0].stats_host()[0]
    identical = all(np.array_equal(G0, s.stats_host()[0]) for s in st.states[1:])
    trace_err = abs(float(np.trace(G0)) - N) / N
    del G0
    st.release()
    slm._state = None
    world = len(devices)
    fl_stats = flops_per_row(d, n)
    fl_pass2 = 2.0 * F * F + 4.0 * d * n
    perr = (None, None)
    if not args.no_parity_check:
        perr = _elbo_parity(make_basis, X, y, var, reg, ls, devices=devices)
        parity("-ELBO of 256 rows vs oracle", perr[0], 1e-4)
        parity("gradient of 256 rows vs oracle (normwise)", perr[1], 2e-3)
    parity("trace(G) / N - 1", trace_err, 1e-6)
    if not identical:
        raise ParityError("the members' reduced statistics differ")
    return {"workload": "StandardLinearModel(devices=%d)._elbo, RandomRBF F=4096 D=32 ARD, N=%d sharded in ONE process" % (world, N),
            "rows": N, "dtype": "f32", "ms": t_eval, "value": N / (t_eval * 1e-3), "unit": "rows/s per _elbo",
            "stage_ms": {"statistics_and_exchange": t_stats, "posterior": t_post, "second_pass": t_pass2},
            "_neg_elbo": float(res[0]), "transport": group.transport,
            "parity": {"neg_elbo_256_rows": perr[0], "gradient_256_rows": perr[1], "trace_rel_err": trace_err,
                       "members_bit_identical": bool(identical)},
            "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS * world,
                         "frac": (fl_stats + fl_pass2) * N / (t_eval * 1e-3) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)}}


def _glm_rows(args):
    """Rows of config 5's data in the N > 1 configurations: BASELINE's 2 M, fewer when --dist-rows asks for a rehearsal."""
    return 2_000_000 if args.dist_rows >= 1_000_000 else max(300_000, 4 * int(args.dist_rows))


def single_process_glm(args, devices, group):
    """Config 5's SVI step with the loop resident on EVERY member of the device group (`GeneralizedLinearModel(devices=...)`:
    rr_glm_sgd_group_step -- each member steps its share of the 65 536-row minibatch, two all-reduces per step in HBM,
    parameters replicated), next to the same fit on ONE context of this process: the interval between the moments the loop
    queues its steps (the queue is two deep: it follows the devices) over steps 8.. of a 56-step fit, device sampler (no
    host generator in the way); parity: the two fits' parameters after 8 steps."""
    import logging
    import revrand_amd.basis_functions as bs
    from revrand_amd import likelihoods as lk
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.glm import GeneralizedLinearModel
    logging.getLogger("revrand_amd").setLevel(logging.ERROR)
    N, d, n, K, L, M = _glm_rows(args), 32, 1024, 10, 50, 65536
    rng = np.random.default_rng([20260928, 5])
    X = rng.standard_normal((N, d), dtype=np.float32)
    y = rng.poisson(np.exp(0.3 * X[:, 0].astype(np.float64))).astype(np.float64)

    def fit(iters, dv):
        g = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                                   K=K, nsamples=L, batch_size=M, maxiter=iters, nstarts=0, random_state=2, sampler="device", devices=dv)
        np.random.seed(20260930)
        g.fit(X, y)
        ck = g.__dict__.get("_resident_clock")
        flat = np.concatenate((g.weights_.ravel(), g.covariance_.ravel(), np.atleast_1d(g.basis_hypers_)))
        return flat, (1e3 * np.diff(ck[8:-1]) if ck is not None and len(ck) > 12 else None)
    out = {}
    p1, _ = fit(8, None)
    pg, _ = fit(8, devices)
    err = float(np.linalg.norm(pg - p1) / np.linalg.norm(p1))
    _, dt1 = fit(56, None)
    _, dtg = fit(56, devices)
    parity("C5 group loop: parameters after 8 steps vs one context (normwise)", err, 1e-4)
    gemm_flops = 3 * 2.0 * K * L * M * 2 * n
    world = len(devices)
    ms1, msg = float(np.median(dt1)), float(np.median(dtg))
    out = {"workload": "GLM Poisson, RandomRBF F=2048 D=32 ARD, N=%d, K=10 L=50, minibatch 65536: SVI step of fit(devices=%d), device sampler" % (N, world),
           "ms": msg, "one_context_ms": ms1, "speedup_vs_one_context": ms1 / msg, "value": M / (msg * 1e-3), "unit": "minibatch-rows/s",
           "dtype": "f32", "transport": group.transport, "parity": {"params_after_8_steps_vs_one_context": err},
           "roofline": {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS * world,
                        "frac": gemm_flops / (msg * 1e-3) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)}}
    return out


def single_process_under_ranks(args, comm, rank, world):
    """Inside the one-process-per-GPU run: once the ranks are done, rank 0 starts `bench.py --gpus N --single-process` as a
    CHILD (the same N GPUs behind one process: the in-process device group of StandardLinearModel(devices=N)) and reports
    its numbers next to the ranks' own -- one driver command then measures both ways of using the node.  The other ranks
    wait on the host (a file, not a collective: a GPU spinning in an RCCL barrier would disturb the measurement); a child
    that fails or hangs costs this entry only."""
    child_limit = min(args.config_timeout, 420.0)  # the child needs ~1 minute at BASELINE's sizes; a stuck one must not cost the line
    tag = "rr_bench_sp_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
    flag = os.path.join(tempfile.gettempdir(), tag)
    comm.barrier()  # every rank has freed its buffers and is idle
    res = None
    if rank == 0:
        try:
            os.unlink(flag)
        except OSError:
            pass
        env = {k: v for k, v in os.environ.items() if k not in (
            "RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "NCCL_HOSTID", "RR_COMM_RDZV", "REVRAND_HIP_DEVICE",
            "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
        if env.get("OMP_NUM_THREADS") == "1":  # torch.distributed.run's per-rank default; the child is ONE process for the node
            env.pop("OMP_NUM_THREADS")
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--single-process", "--steps", str(min(args.steps, 5)),
               "--warmup", str(min(args.warmup, 2)), "--rows", str(args.rows), "--dist-rows", str(args.dist_rows),
               "--config-timeout", str(args.config_timeout)]
        if args.no_parity_check:
            cmd.append("--no-parity-check")
        try:
            p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=child_limit)
            lines = [l for l in p.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                res = {"error": "child exited %d: %s" % (p.returncode, p.stderr.decode(errors="replace")[-600:])}
            else:
                d = json.loads(lines[-1])
                ex, el = d.get("exchange", {}), (d.get("configs") or {}).get("elbo_rbf_f4096_single_process", {})
                gl = (d.get("configs") or {}).get("C5_glm_svi_step_single_process", {})
                res = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                       "n_gpus": d["n_gpus"], "frac": d["roofline"].get("whole_path_frac"),
                       "exchange": {k: ex.get(k) for k in ("transport", "ms_per_step_pack_allreduce_unpack", "busbw_GBps",
                                                           "distinct_gpus", "oversubscribed", "peer_access")},
                       "preflight_busbw_GBps": (ex.get("preflight") or {}).get("busbw_GBps"),
                       "preflight_peer_transport": {k: (ex.get("preflight_peer_transport") or {}).get(k) for k in ("busbw_GBps", "ms", "sum_exact", "error")
                                                    if k in (ex.get("preflight_peer_transport") or {})} or None,
                       "speedup_model": (d.get("per_rank") or {}).get("expected_speedup_model", {}).get("speedup_vs_one_gpu"),
                       "members_bit_identical": d["config"].get("members_bit_identical"),
                       "elbo": {k: el.get(k) for k in ("ms", "stage_ms", "parity", "error") if k in el},
                       "glm_c5": {k: gl.get(k) for k in ("ms", "one_context_ms", "speedup_vs_one_context", "parity", "error") if k in gl}}
        except subprocess.TimeoutExpired:
            res = {"error": "child still running after %.0f s" % child_limit}
        except Exception as e:  # noqa: BLE001
            res = {"error": "%s: %s" % (type(e).__name__, e)}
        with open(flag, "w") as f:
            f.write("done\n")
    else:
        deadline = time.time() + child_limit + 60
        while not os.path.exists(flag) and time.time() < deadline:
            time.sleep(0.2)
    comm.barrier()
    if rank == 0:
        try:
            os.unlink(flag)
        except OSError:
            pass
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=10_000_000, help="global N")
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--nbases", type=int, default=2048)
    ap.add_argument("--cpu-sample", type=int, default=200000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--engine", choices=["f32", "bf16x3", "bf16x4", "fp16x3"], default=None,
                    help="arithmetic of the Gram (default: f32 MFMA, or $RR_SYRK_ENGINE); see DESIGN.md 3.13")
    ap.add_argument("--no-alt-engine", action="store_true", help="skip the informational fp16x3 measurement")
    ap.add_argument("--no-parity-check", action="store_true",
                    help="skip the 2048-row oracle check before timing (profiling runs: keeps per-kernel averages clean)")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("RR_BENCH_LAUNCH_TIMEOUT", "3000")),
                    help="seconds after which the self-launcher (--gpus N without a launcher) ends all ranks and fails")
    ap.add_argument("--config-timeout", type=float, default=900.0,
                    help="seconds one side configuration may take before every thread's stack is dumped and bench.py exits")
    ap.add_argument("--configs", default="all",
                    help="BASELINE's other configurations to time after the headline: all | none | e.g. c3,c5,headline (N=1); "
                         "c3,elbo (N>1: the row-sharded _elbo evaluations)")
    ap.add_argument("--dist-rows", type=int, default=10_000_000,
                    help="global N of the N>1 configurations (BASELINE config 3: 10M; rehearsals on one GPU pass fewer)")
    ap.add_argument("--dist-rows-c4", type=int, default=None,
                    help="global N of the row-sharded config 4 `_elbo` (FastFoodRBF F = 16384; BASELINE: 4 194 304; default: that, "
                         "scaled down with --dist-rows)")
    ap.add_argument("--full-json", default=os.environ.get("RR_BENCH_FULL_JSON"),
                    help="where rank 0 writes the unabridged record (default: gpurun_out/bench_full_n<N>.json when that "
                         "directory exists)")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs behind ONE process: the in-process device group of StandardLinearModel(devices=N) "
                         "(revrand_amd/multigpu.py) instead of one rank per GPU; same line shape")
    args = ap.parse_args()
    if args.dist_rows_c4 is None:
        args.dist_rows_c4 = 4_194_304 if args.dist_rows >= 10_000_000 else max(4096, int(args.dist_rows * 0.4194304))
    faulthandler.enable()  # a crash inside a library call leaves the Python stack on stderr

    if args.single_process:
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)  # RCCL's banner and warnings go to stderr: stdout carries the ONE line
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"
        return single_process(args, json_out)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch(args, sys.argv[1:])

    # stdout carries exactly ONE line (the JSON).  RCCL writes its version banner and warnings to file descriptor 1
    # from its own threads, so fd 1 is pointed at stderr for the whole run and the JSON goes to a private duplicate.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        args.gpus = world
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"  # keep RCCL's version banner out of the logs

    from revrand_amd import _hip, parallel
    if world > 1 and "RR_BENCH_VISIBLE_GPUS" not in os.environ:
        # started by a launcher (torch.distributed.run) on a box with fewer GPUs than ranks -- a rehearsal: the same mapping
        # and host ids as bench.py's own launcher gives (rank r on device r % visible GPUs; RCCL refuses two ranks on one
        # device of one host).  On a node with a GPU per rank nothing changes.
        nvis = _hip.ctypes.c_int()
        ndev = nvis.value if _hip.load_library().rr_device_count(_hip.ctypes.byref(nvis)) == 0 else 0
        if 0 < ndev < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            local_rank %= ndev
            os.environ["LOCAL_RANK"] = str(local_rank)  # what revrand_amd's default device follows
            os.environ["NCCL_HOSTID"] = "rr-bench-rank-%d" % rank
        if ndev > 0:
            os.environ["RR_BENCH_VISIBLE_GPUS"] = str(ndev)
    dev = _hip.get_device(local_rank)
    use_comm = world > 1 or os.environ.get("RR_BENCH_FORCE_DIST") == "1"  # the latter: plumbing check at N=1
    comm = parallel.init_rccl_from_env(device=local_rank) if use_comm else parallel.SingleComm()
    parallel.set_comm(comm)
    rank, world = comm.rank, comm.world  # as RCCL reports them

    # ---- preflight (N > 1): where every rank's GPU sits, whether the ranks have a GPU each, and config 3's 273 MB message
    # through the communicator BEFORE anything is timed -- the first collective between two physical devices must not sit
    # in a timed region, and its bus bandwidth tells xGMI / PCIe / sockets apart at a glance ----
    place, pre, placements = placement(dev, pin=world > 1), None, None  # (N = 1: the cpu_baseline keeps all host cores)
    if use_comm:
        def pci_numbers(s):
            try:
                dom, bus, rest = s.split(":")
                dv, fn = rest.split(".")
                return [int(dom, 16), int(bus, 16), int(dv, 16), int(fn, 16)]
            except Exception:
                return [-1, -1, -1, -1]
        row = np.zeros((world, 7))
        row[rank] = pci_numbers(place.get("pci", "")) + [place.get("numa") if place.get("numa") is not None else -1,
                                                         place.get("pinned_cpus", 0), local_rank]
        table = comm.allreduce_host(row.ravel()).reshape(world, 7).astype(int)
        placements = [{"rank": r, "pci": "%04x:%02x:%02x.%x" % tuple(t[:4]) if t[0] >= 0 else None, "numa": int(t[4]) if t[4] >= 0 else None,
                       "pinned_cpus": int(t[5]), "device": int(t[6])} for r, t in enumerate(table)]
        cnt3 = parallel.stats_count(8257)
        pre_buf = dev.zeros(cnt3 * 8)
        pre = preflight_exchange(lambda: comm.allreduce_device(pre_buf, cnt3), dev.sync, dev, cnt3, world)
        pre_buf.free()
        pre["ms"] = float(comm.allreduce_host(np.array([pre["ms"]]), op="max")[0])  # the slowest rank's
        if world > 1 and pre["ms"] > 0:
            pre["busbw_GBps"] = 2.0 * (world - 1) / world * pre["message_bytes"] / (pre["ms"] * 1e-3) / 1e9
            pre["busbw_frac_of_one_link"] = pre["busbw_GBps"] / XGMI_LINK_GBPS

    d, n = args.dim, args.nbases
    F = 2 * n
    W = np.random.RandomState(42).randn(d, n)
    wvec = np.random.RandomState(1).randn(d).astype(np.float32)
    basis = _hip.RffHandle(W, compute="f32", device=local_rank)
    if args.engine:
        dev.set_gram_engine(args.engine)
    engine = dev.gram_engine

    # ---- this rank's contiguous row shard (equal to within one row), generated in 250k-row chunks
    # (chunk c of the data set always comes from RNG stream c, so the data do not depend on N_gpus)
    # and made resident in HBM before anything is timed ----
    CH = 250_000
    row0, row1 = parallel.shard_bounds(args.rows, rank, world)
    my_rows = row1 - row0
    dX = dev.empty_matrix(my_rows, d, np.float32, ld_dev=basis.padded_dim)
    dy = dev.malloc(max(my_rows, 1) * 4)
    dy.dtype = np.dtype(np.float32)
    r0 = 0
    for c in range(row0 // CH, (row1 + CH - 1) // CH if my_rows else 0):
        c0 = c * CH
        Xc, yc = gen_chunk(c, min(CH, args.rows - c0), d, wvec)
        lo, hi = max(row0, c0) - c0, min(row1, c0 + CH) - c0
        Xc, yc = np.ascontiguousarray(Xc[lo:hi]), np.ascontiguousarray(yc[lo:hi])
        dev.upload_rows(dX, r0, Xc)
        _hip._check(dev.lib, dev.lib.rr_memcpy_h2d(dev.ctx, _hip.ctypes.c_void_p(dy.ptr.value + r0 * 4),
                                                   yc.ctypes.data_as(_hip.ctypes.c_void_p), yc.nbytes))
        r0 += hi - lo
    assert r0 == my_rows

    # ---- accumulators: [G (F*F) | b (F) | yty (1)] float64 ----
    nacc = F * F + F + 1
    acc_buf = dev.zeros(nacc * 8)
    acc_ptr = acc_buf.ptr.value
    pG = _hip.ctypes.c_void_p(acc_ptr)
    pb = _hip.ctypes.c_void_p(acc_ptr + F * F * 8)
    pt = _hip.ctypes.c_void_p(acc_ptr + (F * F + F) * 8)
    kernel_ms = []
    exch_ms = []

    def step(timed):
        _hip._check(dev.lib, dev.lib.rr_memset(dev.ctx, pG, 0, nacc * 8))
        if my_rows:
            basis.gram_dev(dX, dy, 1.0, pG, pb, pt)
            if timed:
                kernel_ms.append(basis.gram_timings())  # HIP events on the kernels' own stream
        if use_comm:
            # the one exchange step of the path: pack the upper triangle, ncclAllReduce over xGMI, unpack + mirror --
            # all on the context's stream behind the Gram kernels
            if timed:
                dev.timer_start()
            comm.reduce_stats_device(F, pG, pb, pt, my_rows, wait=False)
            if timed:
                exch_ms.append(dev.timer_stop())
        else:
            _hip._check(dev.lib, dev.lib.rr_symmetrize_dev(dev.ctx, pG, F))
        dev.sync()

    def barrier():
        dev.sync()
        comm.barrier()

    # ---- parity of the measured path on a slice, off-diagonal entries included: Gram of the first rows against the
    # oracle (checker only) before anything is timed ----
    parity_err = None
    if rank == 0 and my_rows >= 2048 and not args.no_parity_check:
        orc = _oracle()
        Xs, ys = gen_chunk(row0 // CH, min(CH, args.rows - (row0 // CH) * CH), d, wvec)
        lo = row0 - (row0 // CH) * CH
        Xs, ys = Xs[lo:lo + 2048], ys[lo:lo + 2048]
        if len(Xs) == 2048:
            dXs = _hip.DeviceMatrix(dev, _hip.ctypes.c_void_p(dX.ptr.value), (2048, d), dX.ld, np.float32)
            dys = _hip.DeviceBuffer(dev, _hip.ctypes.c_void_p(dy.ptr.value), 2048 * 4)
            dys.dtype = np.dtype(np.float32)
            _hip._check(dev.lib, dev.lib.rr_memset(dev.ctx, pG, 0, nacc * 8))
            basis.gram_dev(dXs, dys, 1.0, pG, pb, pt)
            dev.sync()
            _hip._check(dev.lib, dev.lib.rr_symmetrize_dev(dev.ctx, pG, F))
            dev.sync()
            dXs.ptr = None
            dys.ptr = None
            Gs = dev.download(acc_buf, (F, F), np.float64)
            Gr, _, _ = orc.rff_gram_chunked(Xs.astype(np.float64), ys.astype(np.float64), W, 1.0)
            parity_err = float(np.abs(Gs - Gr).max() / np.abs(Gr).max())
            assert parity_err < 1e-4 or os.environ.get("RR_GRAM_ABLATE"), parity_err

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = float(comm.allreduce_host(np.array([elapsed]), op="max")[0])  # MAX over ranks
    # per-rank device time of one step's kernels (HIP events) and of the exchange: max / min / sum over the ranks
    my_k = float(np.mean([k[0] + k[1] + k[2] for k in kernel_ms])) if kernel_ms else 0.0
    my_x = float(np.mean(exch_ms)) if exch_ms else 0.0
    k_max, x_max = comm.allreduce_host(np.array([my_k, my_x]), op="max")
    k_min, x_min = comm.allreduce_host(np.array([my_k, my_x]), op="min")
    k_sum = float(comm.allreduce_host(np.array([my_k]))[0])

    # sanity on the result of the last step (every rank holds the global statistics): trace(G) == N
    # (cos^2 + sin^2 = 1 per frequency), exactly symmetric, summed row count == N
    G = dev.download(acc_buf, (F, F), np.float64)
    diag = float(np.trace(G))
    assert np.array_equal(G, G.T) or os.environ.get("RR_GRAM_ABLATE")
    trace_err = abs(diag - args.rows) / args.rows

    # ---- full-size parity (N = 1): EVERY entry of the timed step's G and b, over all rows, against the same statistics in
    # float64 arithmetic (rr_syrk_f64_kernel over float64 features of float64 X -- the route tests/test_gpu_rff.py holds to the
    # oracle at 1e-5): what the 2048-row oracle slice and the trace cannot see at N = 10M.  Untimed. ----
    full_check = None
    if world == 1 and not use_comm and engine == "f32" and not args.no_parity_check and not os.environ.get("RR_GRAM_ABLATE") \
            and my_rows <= 20_000_000 and os.environ.get("RR_BENCH_NO_FULL_CHECK") != "1":
        t_fc = time.perf_counter()
        b64 = _hip.RffHandle(W, compute="f64", device=local_rank)
        dX64 = dev.empty_matrix(my_rows, d, np.float64, ld_dev=b64.padded_dim)
        dy64 = dev.malloc(max(my_rows, 1) * 8)
        dy64.dtype = np.dtype(np.float64)
        r0 = 0
        for c in range((my_rows + CH - 1) // CH):
            Xc, yc = gen_chunk(c, min(CH, args.rows - c * CH), d, wvec)
            dev.upload_rows(dX64, r0, Xc.astype(np.float64))
            yc = yc.astype(np.float64)
            _hip._check(dev.lib, dev.lib.rr_memcpy_h2d(dev.ctx, _hip.ctypes.c_void_p(dy64.ptr.value + r0 * 8),
                                                       yc.ctypes.data_as(_hip.ctypes.c_void_p), yc.nbytes))
            r0 += len(yc)
        acc64 = dev.zeros(nacc * 8)
        q = acc64.ptr.value
        b64.gram_dev(dX64, dy64, 1.0, _hip.ctypes.c_void_p(q), _hip.ctypes.c_void_p(q + F * F * 8), _hip.ctypes.c_void_p(q + (F * F + F) * 8))
        _hip._check(dev.lib, dev.lib.rr_symmetrize_dev(dev.ctx, _hip.ctypes.c_void_p(q), F))
        dev.sync()
        G64 = dev.download(acc64, (F, F), np.float64)
        bv = dev.download(acc_buf, (F,), np.float64, offset_bytes=F * F * 8)
        bv64 = dev.download(acc64, (F,), np.float64, offset_bytes=F * F * 8)
        full_check = {"rows": my_rows, "G_max_rel_diff_f32_vs_f64": float(np.abs(G - G64).max() / np.abs(G64).max()),
                      "b_max_rel_diff_f32_vs_f64": float(np.abs(bv - bv64).max() / np.abs(bv64).max()),
                      "_seconds": time.perf_counter() - t_fc}
        for buf in (dX64, dy64, acc64):
            buf.free()
        del b64, G64
        assert full_check["G_max_rel_diff_f32_vs_f64"] < 1e-4 and full_check["b_max_rel_diff_f32_vs_f64"] < 1e-4, full_check

    # ---- informational: the same step on the split-fp16 engine (N=1 only; never `value`) ----
    alt = None
    if world == 1 and not use_comm and engine == "f32" and not args.no_alt_engine and not os.environ.get("RR_GRAM_ABLATE"):
        dev.set_gram_engine("fp16x3")
        step(False)
        dev.sync()
        ta = time.perf_counter()
        nalt = min(max(args.steps, 1), 2)
        alt_ms = []
        for _ in range(nalt):
            step(False)
            alt_ms.append(basis.gram_timings())
        dev.sync()
        alt_elapsed = (time.perf_counter() - ta) / nalt
        G3 = dev.download(acc_buf, (F, F), np.float64)
        kms_alt = float(np.mean([k[1] + k[2] for k in alt_ms]))
        nbk = len(range(0, F, 256))
        # 136 full 256x256 tiles x 3 products are issued for F (F + 1) algorithmic flops per row
        issued = 3.0 * 2.0 * 256 * 256 * (nbk * (nbk + 1) // 2) * my_rows / (kms_alt * 1e-3) / 1e12
        alt = {"engine": "fp16x3", "value": args.rows / alt_elapsed, "ms_per_step": 1e3 * alt_elapsed, "steps": nalt,
               "max_diff_vs_f32_engine": float(np.abs(G3 - G).max() / np.abs(G).max()),
               "trace_rel_err": abs(float(np.trace(G3)) - args.rows) / args.rows,
               "kernel_ms_per_step": kms_alt, "mfma_issued_frac_of_fp16_peak": issued / PEAK_BF16_MFMA_TFLOPS,
               "algorithmic_tflops": F * (F + 1.0) * my_rows / (kms_alt * 1e-3) / 1e12,
               "_kernel": "rr_syrk_b16w4_kernel<3, false, true>", "_mfma_issued_tflops": issued,
               "_features_ms_per_step": float(np.mean([k[0] for k in alt_ms])),
               "_what": "opt-in engine (Device.set_gram_engine / RR_SYRK_ENGINE), DESIGN.md 3.13: features scaled into [-1, 1] "
                        "and split into fp16 hi + lo, 3 products on the fp16 matrix pipe, f32 accumulation"}
        dev.set_gram_engine("f32")
    del G

    out = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        value = args.rows / (elapsed / max(args.steps, 1))
        # Kernels of one step: features (projection + cos/sin + Phi^T y), off-diagonal SYRK (dominant),
        # diagonal SYRK.  Algorithmic work per row (upper triangle incl. diagonal, 2 flop per
        # entry): off-diagonal tiles 2 sum_{i<j} w_i w_j, diagonal tiles sum_i w_i (w_i + 1), w_i = valid
        # columns of 256-column block i; the projection 2dn and Phi^T y 2F belong to the features kernel.
        feat_ms = float(np.mean([k[0] for k in kernel_ms])) if kernel_ms else float("nan")
        syrk_ms = float(np.mean([k[1] for k in kernel_ms])) if kernel_ms else float("nan")
        diag_ms = float(np.mean([k[2] for k in kernel_ms])) if kernel_ms else float("nan")
        launches = kernel_ms[0][3] if kernel_ms else 0
        widths = [min(256, F - 256 * i) for i in range((F + 255) // 256)]
        off_flops = 2.0 * sum(widths[i] * widths[j] for i in range(len(widths)) for j in range(i + 1, len(widths)))
        diag_flops = float(sum(w * (w + 1) for w in widths))
        assert off_flops + diag_flops == F * (F + 1.0)
        achieved = off_flops * my_rows / (syrk_ms * 1e-3) / 1e12 if kernel_ms else float("nan")
        gram_tf = (off_flops + diag_flops) * my_rows / ((syrk_ms + diag_ms) * 1e-3) / 1e12 if kernel_ms else float("nan")
        rt = runtime_info(_hip, parallel)
        out = {
            "metric": "feature-rows/sec (Phi + PhiT Phi + PhiT y) at N=%s D=%d F=%d" % (
                "10M" if args.rows == 10_000_000 else args.rows, d, F),
            "value": value, "unit": "feature-rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RandomRBF nbases=%d (F=%d), D=%d, N=%d f32, features + MFMA Gram, rows sharded "
                                   "over %d GPU(s)" % (n, F, d, args.rows, world),
                       "rows_per_gpu": my_rows, "device": dev.name.strip(), "trace_rel_err": trace_err,
                       "gram_engine": engine, "parity_rel_err_2048_rows_vs_oracle": parity_err,
                       "parity_all_rows_f32_vs_f64": full_check,
                       "runtime": {"hip_runtime": os.path.basename(rt["hip_runtime"] or ""),
                                   "rccl": rt["rccl"].get("version", rt["rccl"].get("error"))},
                       "_runtime": rt},
            "roofline": {"bound": "mfma", "kernel": "rr_syrk_f32_kernel", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         # bytes per launch: the profiled launch's bytes/row x this run's rows per launch
                         "traffic": (TRAFFIC["hbm_bytes"] / TRAFFIC["rows_per_launch"] * my_rows / max(launches, 1)
                                     if TRAFFIC else None),
                         # (not observed in this run: counters need their own rocprofv3 --pmc passes)
                         "traffic_source": "profiles/traffic.json" if TRAFFIC else None,
                         "_traffic_note": TRAFFIC.get("note"),
                         "kernel_ms_per_step": syrk_ms, "launches_per_step": launches,
                         "avg_launch_ms": syrk_ms / max(launches, 1),
                         "flops_per_row": off_flops, "rows_per_step": my_rows,
                         "other_kernels_ms_per_step": {"rr_syrk_f32_diag16_kernel<32>": diag_ms,
                                                       "rr_rff_features_mfma_kernel": feat_ms},
                         "gram_both_kernels_frac": gram_tf / PEAK_F32_MFMA_TFLOPS,
                         "whole_path_frac": flops_per_row(d, n) * args.rows / (elapsed / max(args.steps, 1))
                         / world / 1e12 / PEAK_F32_MFMA_TFLOPS},
        }
        if use_comm:
            cnt = parallel.stats_count(F)
            xm = float(np.mean(exch_ms)) if exch_ms else None
            out["exchange"] = {"transport": "RCCL ncclAllReduce(f64, sum), bound directly (rr_comm_*)",
                               "rccl": dict(zip(("version", "library"), parallel.RcclComm.load())),
                               "ranks_rccl_reports": world, "message_float64": cnt, "message_bytes": 8 * cnt,
                               "ms_per_step_pack_allreduce_unpack": xm,
                               "ms_per_step_pack_allreduce_unpack_max_min_over_ranks": [float(x_max), float(x_min)],
                               "busbw_GBps": 2.0 * (world - 1) / world * 8 * cnt / (float(x_max) * 1e-3) / 1e9 if x_max > 0 else None,
                               "model_ms_at_one_xgmi_link": exchange_model_ms(8 * cnt, world),
                               "visible_gpus": int(os.environ.get("RR_BENCH_VISIBLE_GPUS", "0")) or None,
                               "distinct_gpus": len({p["pci"] for p in placements}) if placements else None,
                               "oversubscribed": bool(os.environ.get("NCCL_HOSTID", "").startswith("rr-bench-rank-")),
                               "preflight": pre, "placement": placements}
        if world > 1:
            # what the ranks' own clocks say: device time of a step's kernels on the slowest / fastest rank, and the model
            # "largest shard's kernels + exchange" against the same kernels run back to back on one GPU (their sum)
            model_ms = float(k_max) + float(x_max)
            out["per_rank"] = {"kernel_ms_per_step_max_min_over_ranks": [float(k_max), float(k_min)],
                               "kernel_ms_per_step_sum_over_ranks": k_sum,
                               "expected_speedup_model": {"ms_per_step": model_ms,
                                                          "speedup_vs_one_gpu": k_sum / model_ms if model_ms > 0 else None,
                                                          "measured_ms_per_step": ms_per_step,
                                                          "_what": "max over ranks of (features + SYRK kernels) + exchange; "
                                                                   "one GPU = the same kernels of all shards back to back"}}
        if engine != "f32":
            # split-bf16 engine: one SYRK kernel over all 136 tiles; the roofline is the bf16 matrix pipe, `achieved`
            # stays ALGORITHMIC flops (the kernel issues 3 or 4 bf16 products per f32 product: `issued_frac`)
            nprod = 4 if engine == "bf16x4" else 3
            k_ms = syrk_ms + diag_ms
            nbk = len(widths)
            alg = (off_flops + diag_flops) * my_rows / (k_ms * 1e-3) / 1e12
            issued = nprod * 2.0 * 65536 * (nbk * (nbk + 1) // 2) * my_rows / (k_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "rr_syrk_b16w4_kernel<%d, ...>" % nprod, "achieved": alg,
                               "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": alg / PEAK_BF16_MFMA_TFLOPS,
                               "issued": issued, "issued_frac": issued / PEAK_BF16_MFMA_TFLOPS, "traffic": None,
                               "kernel_ms_per_step": k_ms, "launches_per_step": launches,
                               "avg_launch_ms": k_ms / max(launches, 1), "flops_per_row": off_flops + diag_flops,
                               "rows_per_step": my_rows,
                               "other_kernels_ms_per_step": {"rr_rff_features_mfma_kernel": feat_ms},
                               "f32_mfma_equivalent_frac": alg / PEAK_F32_MFMA_TFLOPS}
            out["dtype"] = "f32 values as %s hi+lo, %d 16-bit products per f32 product, f32 accumulate" % (
                "fp16" if engine == "fp16x3" else "bf16", nprod)
        if alt:
            out["split_fp16_engine"] = alt
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d, n, W, wvec, args.cpu_sample)
    ok = trace_err < (1e-6 if engine == "f32" else 1e-5) or os.environ.get("RR_GRAM_ABLATE")

    def emit(configs=None):
        """Rank 0: THE line (numbers, < 8 KB) on stdout; the unabridged record to --full-json."""
        if rank != 0:
            return
        if configs is not None:
            out["configs"] = configs
        line = json.dumps(lean(out), separators=(",", ":"))
        if len(line) > 8000:
            sys.stderr.write("bench.py: WARNING: the JSON line has %d bytes (driver keeps a 9 KB tail)\n" % len(line))
        json_out.write(line + "\n")
        json_out.flush()
        path = args.full_json
        if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            path = os.path.join(ROOT, "gpurun_out", "bench_full_n%d.json" % world)
        if path:
            try:
                with open(path, "w") as f:
                    json.dump(full(out), f, indent=1)
            except OSError as e:
                sys.stderr.write("bench.py: could not write %s: %s\n" % (path, e))

    side = args.configs != "none" and engine == "f32" and ok and not os.environ.get("RR_GRAM_ABLATE")
    if side and (world > 1 or rank == 0):
        # free the headline's buffers first: C3 / C4 want tens of GB
        for b in (dX, dy, acc_buf):
            b.free()
        del basis
    if side and world == 1 and not use_comm:
        emit(extra_configs(dev, _hip, args, emit))
    elif side and world > 1:
        cfgs = dist_configs(dev, _hip, comm, args, emit)
        if isinstance(cfgs, dict) and os.environ.get("RR_BENCH_NO_SINGLE_PROCESS") != "1":
            cfgs["single_process"] = single_process_under_ranks(args, comm, rank, world)
        emit(cfgs)
    else:
        emit()
    comm.barrier()
    comm.close()
    assert ok, trace_err


if __name__ == "__main__":
    main()
