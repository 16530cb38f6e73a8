#!/usr/bin/env python3
"""
bench.py -- feature-rows/sec of the hot path (Phi + Phi^T Phi + Phi^T y) on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): RandomRBF, N = 10M rows, D = 32, F = 2*nbases = 4096,
f32 arithmetic, synthetic Gaussian inputs resident in HBM before the timed region.  One
"step" is one full pass of the fused kernel over this rank's rows: zero the (F,F)/(F,)
accumulators, project + cos/sin + accumulate every row, (N>1: all-reduce the partial Gram
over RCCL), mirror the triangle.  N>1 runs one process per GPU (torch.distributed.run sets
RANK/LOCAL_RANK/WORLD_SIZE); rows are sharded across ranks (fixed global N -> "strong").

Rank 0 prints ONE JSON line with the contract's keys plus
  roofline     -- algorithmic flops of one launch / HIP-event time of the kernel, vs the
                  f32 MFMA peak of gfx950 (157.3 TFLOP/s; MI355X_MICROARCH.md)
  cpu_baseline -- the NumPy restatement of revrand's path (oracle/, kind "port") timed on
                  this box's host cores over a bounded row sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA", dense (same rate for fp16)

# HBM-side traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (tools_prof.sh),
# corrected as MI355X_MICROARCH.md prescribes; the committed summary is quoted, per launch.
TRAFFIC = {}
try:
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as _f:
        TRAFFIC = json.load(_f)
except Exception:
    pass


def flops_per_row(d, n):
    F = 2 * n
    return 2.0 * d * n + F * (F + 1.0) + 2.0 * F  # projection + upper-tri Gram + Phi^T y


def gen_chunk(c, rows, d, wvec):
    """Chunk c of the synthetic data set: X ~ N(0,1) f32, y = sin(X w) + 0.1 eps."""
    rng = np.random.default_rng([20260928, c])
    X = rng.standard_normal((rows, d), dtype=np.float32)
    y = np.sin(X @ wvec) + 0.1 * rng.standard_normal(rows, dtype=np.float32)
    return X, y.astype(np.float32)


def cpu_baseline(d, n, W, wvec, sample_rows):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import revrand_oracle as orc  # checker/baseline only -- never on the product path
    X, y = gen_chunk(10 ** 6, sample_rows, d, wvec)
    X64, y64 = X.astype(np.float64), y.astype(np.float64)
    orc.rff_gram_chunked(X64[:2000], y64[:2000], W, 1.0, chunk=1000)  # warm BLAS
    t0 = time.perf_counter()
    orc.rff_gram_chunked(X64, y64, W, 1.0, chunk=10000)
    dt = time.perf_counter() - t0
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        nt = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        threads = max(nt) if nt else threads
    except Exception:
        pass
    return {"value": sample_rows / dt, "unit": "feature-rows/s", "cores": int(threads), "kind": "port",
            "sample": "%d rows of the same workload, f64, 10000-row chunks, %.1f s" % (sample_rows, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=10_000_000, help="global N")
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--nbases", type=int, default=2048)
    ap.add_argument("--cpu-sample", type=int, default=150000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--engine", choices=["f32", "bf16x3", "bf16x4", "fp16x3"], default=None,
                    help="arithmetic of the Gram (default: f32 MFMA, or $RR_SYRK_ENGINE); see DESIGN.md 3.13")
    ap.add_argument("--no-alt-engine", action="store_true", help="skip the informational fp16x3 measurement")
    ap.add_argument("--no-parity-check", action="store_true",
                    help="skip the 2048-row oracle check before timing (profiling runs: keeps per-kernel averages clean)")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON).  RCCL writes its version banner and warnings to file descriptor 1
    # from its own threads, so fd 1 is pointed at stderr for the whole run and the JSON goes to a private duplicate.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch --gpus %d with: python -m torch.distributed.run --nnodes=1 "
                             "--nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ..." %
                             (args.gpus, args.gpus, args.gpus))
        args.gpus = world

    dist = None
    torch = None
    use_dist = world > 1 or os.environ.get("RR_BENCH_FORCE_DIST") == "1"  # the latter: plumbing check at N=1
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"  # keep RCCL's version banner off stdout (one JSON line)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from revrand_amd import _hip, parallel
    dev = _hip.get_device(local_rank)
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

    # ---- accumulators: [G (F*F) | b (F) | yty (1)] float64, one buffer so one all-reduce ----
    nacc = F * F + F + 1
    if use_dist:
        acc_t = torch.zeros(nacc, dtype=torch.float64, device="cuda:%d" % local_rank)
        acc_ptr = acc_t.data_ptr()
        torch.cuda.synchronize()
    else:
        acc_buf = dev.zeros(nacc * 8)
        acc_ptr = acc_buf.ptr.value
    pG = _hip.ctypes.c_void_p(acc_ptr)
    pb = _hip.ctypes.c_void_p(acc_ptr + F * F * 8)
    pt = _hip.ctypes.c_void_p(acc_ptr + (F * F + F) * 8)
    kernel_ms = []

    def step(timed):
        _hip._check(dev.lib, dev.lib.rr_memset(dev.ctx, pG, 0, nacc * 8))
        if my_rows:
            basis.gram_dev(dX, dy, 1.0, pG, pb, pt)
            if timed:
                kernel_ms.append(basis.gram_timings())  # HIP events on the kernels' own stream
        dev.sync()
        if use_dist:
            if world > 1:
                parallel.allreduce_packed(acc_t)  # RCCL over xGMI: the one exchange step of the path
            else:
                dist.all_reduce(acc_t)  # N=1 plumbing check only
            torch.cuda.synchronize()
        _hip._check(dev.lib, dev.lib.rr_symmetrize_dev(dev.ctx, pG, F))
        dev.sync()

    def barrier():
        dev.sync()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()

    # ---- parity of the measured path on a slice, off-diagonal entries included: Gram of the first rows against the
    # oracle (checker only) before anything is timed ----
    parity_err = None
    if rank == 0 and my_rows >= 2048 and not args.no_parity_check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import revrand_oracle as orc
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
            if use_dist:
                Gs = acc_t[:F * F].view(F, F).cpu().numpy()
            else:
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
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda:%d" % local_rank)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity on the result of the last step: trace(G) == N (cos^2 + sin^2 = 1 per frequency)
    if use_dist:
        diag = acc_t[:F * F].view(F, F).diagonal().sum().item()
    else:
        G = dev.download(acc_buf, (F, F), np.float64)
        diag = float(np.trace(G))
        assert np.array_equal(G, G.T) or os.environ.get("RR_GRAM_ABLATE")
    trace_err = abs(diag - args.rows) / args.rows

    # ---- informational: the same step on the split-fp16 engine (N=1 only; never `value`) ----
    alt = None
    if world == 1 and not use_dist and engine == "f32" and not args.no_alt_engine and not os.environ.get("RR_GRAM_ABLATE"):
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
        Gr_alt = None
        alt = {"engine": "fp16x3", "value": args.rows / alt_elapsed, "unit": "feature-rows/s",
               "ms_per_step": 1e3 * alt_elapsed, "steps": nalt,
               "max_abs_diff_vs_f32_engine_over_max_G": float(np.abs(G3 - G).max() / np.abs(G).max()),
               "kernel": "rr_syrk_b16w4_kernel<3, false, true>",
               "kernel_ms_per_step": float(np.mean([k[1] + k[2] for k in alt_ms])),
               "features_ms_per_step": float(np.mean([k[0] for k in alt_ms])),
               "trace_rel_err": abs(float(np.trace(G3)) - args.rows) / args.rows,
               "note": "features scaled into [-1, 1] and split into fp16 hi + lo (22 mantissa bits), 3 products on the "
                       "fp16 matrix pipe, f32 accumulation: same error against the float64 oracle as the f32 MFMA engine "
                       "(tests/test_gpu_gram_engines.py); opt-in (Device.set_gram_engine / RR_SYRK_ENGINE), DESIGN.md 3.13"}
        # 136 full 256x256 tiles x 3 products are issued for F (F + 1) algorithmic flops per row
        alt["mfma_issued_tflops"] = 3.0 * 2.0 * 256 * 256 * (len(range(0, F, 256)) * (len(range(0, F, 256)) + 1) // 2) \
            * my_rows / (alt["kernel_ms_per_step"] * 1e-3) / 1e12
        alt["mfma_issued_frac_of_fp16_peak"] = alt["mfma_issued_tflops"] / PEAK_BF16_MFMA_TFLOPS
        alt["algorithmic_tflops"] = F * (F + 1.0) * my_rows / (alt["kernel_ms_per_step"] * 1e-3) / 1e12
        dev.set_gram_engine("f32")

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
        out = {
            "metric": "feature-rows/sec (Phi + PhiT Phi + PhiT y) at N=%s D=%d F=%d" % (
                "10M" if args.rows == 10_000_000 else args.rows, d, F),
            "value": value, "unit": "feature-rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RandomRBF nbases=%d (F=%d), D=%d, N=%d f32, features + MFMA Gram, rows sharded "
                                   "over %d GPU(s)" % (n, F, d, args.rows, world),
                       "rows_per_gpu": my_rows, "device": dev.name, "trace_rel_err": trace_err,
                       "gram_engine": engine, "parity_rel_err_2048_rows_vs_oracle": parity_err},
            "roofline": {"bound": "mfma", "kernel": "rr_syrk_f32_kernel", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         # bytes per launch: the profiled launch's bytes/row x this run's rows per launch
                         "traffic": (TRAFFIC["hbm_bytes"] / TRAFFIC["rows_per_launch"] * my_rows / max(launches, 1)
                                     if TRAFFIC else None),
                         "traffic_note": TRAFFIC.get("note"),
                         "kernel_ms_per_step": syrk_ms, "launches_per_step": launches,
                         "avg_launch_ms": syrk_ms / max(launches, 1),
                         "flops_per_row": off_flops, "rows_per_step": my_rows,
                         "other_kernels_ms_per_step": {"rr_syrk_f32_diag_kernel": diag_ms,
                                                       "rr_rff_features_mfma_kernel": feat_ms},
                         "gram_both_kernels_frac": gram_tf / PEAK_F32_MFMA_TFLOPS,
                         "whole_path_frac": flops_per_row(d, n) * args.rows / (elapsed / max(args.steps, 1))
                         / world / 1e12 / PEAK_F32_MFMA_TFLOPS},
        }
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
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if use_dist:
        dist.destroy_process_group()
    assert trace_err < (1e-6 if engine == "f32" else 1e-5) or os.environ.get("RR_GRAM_ABLATE"), trace_err


if __name__ == "__main__":
    main()
