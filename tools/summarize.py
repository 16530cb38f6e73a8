#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a tools/prof.sh run (gpurun_out/prof_<round>) into profiles/<round>_<name>/ and derive a
summary.json per profile.

Every fraction is computed BY TOTALS, so that it can be redone by hand from the committed kernel_stats.csv:

    frac = (algorithmic work per row) x (rows of ALL launches of the kernel) / TotalDurationNs / peak

with rows of all launches = (launches / launches per pass) x rows per pass -- the profiled commands run with
--no-parity-check, so a kernel's launches are the passes' row chunks and nothing else.  Launches are never selected by their
duration (round 3's summary priced a 475 712-row remainder launch as a 524 288-row one and reported 0.995 for a kernel at
0.949).  Where a configuration's kernel also serves another stage (the f64 GEMM: second pass and posterior), launches are
selected by GRID SIZE from the kernel trace, and the selection is written into the summary.

    python tools/summarize.py [round]          (default r06)
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + ROUND)
PEAK = {"f32": 157.3e12, "f64": 78.6e12, "hbm": 8.0e12}
MAX_CLOCK_FRAC = 1.0  # a fraction above clock / 2.40 GHz cannot be right: checked below against 1.0


def bench_record(tag):
    for name in (tag + ".full.json", tag + ".json"):
        p = os.path.join(SRC, name)
        if os.path.exists(p):
            txt = open(p).read()
            if name.endswith(".full.json"):
                return json.loads(txt)
            lines = [l for l in txt.splitlines() if l.startswith("{")]
            if lines:
                return json.loads(lines[-1])
    return None


def kernel_stats(tag):
    p = os.path.join(SRC, tag, "kt_kernel_stats.csv")
    out = {}
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            out[r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                              "total_ms": float(r["TotalDurationNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6,
                              "max_ms": float(r["MaxNs"]) / 1e6}
    return out


def kernel_trace(tag):
    """[(name, ms, grid (work-items: X x Y x Z), workgroup_x)]"""
    p = os.path.join(SRC, tag, "kt_kernel_trace.csv")
    out = []
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            out.append((r["Kernel_Name"].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                        int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"])))
    return out


def by_totals(stats, trace, prefix, work_per_row, rows_per_pass, launches_per_pass, peak, unit="flop", what="", min_grid=None, first_n=None):
    """The kernel's fraction of `peak` over ALL its launches (or, with min_grid, over the launches of at least that grid)."""
    names = [k for k in stats if k.replace("void ", "").startswith(prefix)]
    if not names:
        return None
    if min_grid is None:
        calls = sum(stats[k]["calls"] for k in names)
        total = sum(stats[k]["total_ms"] for k in names)
        lo, hi = min(stats[k]["min_ms"] for k in names), max(stats[k]["max_ms"] for k in names)
        sel = "all launches (kernel_stats.csv: Calls, TotalDurationNs)"
    else:
        grids = [g for n, _, g, _ in trace if n.startswith(prefix)]
        if not grids:
            return None
        exact = False
        if min_grid == "max":   # the kernel's largest launches (the passes'; the posterior's launches of the same kernel are small)
            min_grid = max(grids)
        elif min_grid == "mode":  # the configuration's equal row chunks; the bench's (small) headline part launched the same kernel once
            min_grid, exact = max(set(grids), key=grids.count), True
        d = [ms for n, ms, g, _ in trace if n.startswith(prefix) and (g == min_grid if exact else g >= min_grid)]
        if first_n is not None:  # the configuration's streaming passes come first in the trace; later launches of the same grid belong to another stage
            d = d[:first_n]
        calls, total, lo, hi = len(d), sum(d), min(d), max(d)
        sel = "launches with grid (X x Y x Z work-items) %s %d%s (kernel trace; %d launches of the kernel in all)" % (
            "==" if exact else ">=", min_grid, "" if first_n is None else ", the first %d in launch order" % first_n, len(grids))
    passes = calls / float(launches_per_pass)
    assert abs(passes - round(passes)) < 1e-9, (prefix, calls, launches_per_pass)
    rows = rows_per_pass * passes
    rate = work_per_row * rows / (total * 1e-3)
    frac = rate / peak
    assert frac <= MAX_CLOCK_FRAC, (prefix, frac)
    return {"launches": calls, "launches_selected": sel, "launches_per_pass": launches_per_pass, "passes": passes,
            "rows_per_pass": rows_per_pass, "rows_all_launches": rows, "total_ms": total, "min_ms": lo, "max_ms": hi,
            "algorithmic_%s_per_row" % unit: work_per_row,
            "achieved": rate / (1e12 if unit == "flop" else 1e9), "achieved_unit": "TFLOP/s" if unit == "flop" else "GB/s",
            "frac_of_peak": frac, "what": what}


def put(name, tag, summary):
    dst = os.path.join(ROOT, "profiles", "%s_%s" % (ROUND, name))
    os.makedirs(dst, exist_ok=True)
    for src, dname in ((os.path.join(SRC, tag, "kt_kernel_stats.csv"), "kernel_stats.csv"),
                       (os.path.join(SRC, tag + ".json"), "bench_under_rocprof.json")):
        if os.path.exists(src):
            shutil.copy(src, os.path.join(dst, dname))
    summary["how"] = ("frac_of_peak = algorithmic work per row x rows of all launches / total_ms / peak; rows of all launches = "
                      "launches / launches_per_pass x rows_per_pass (the profiled command ran with --no-parity-check: every launch "
                      "is a pass' row chunk).  Redo it from kernel_stats.csv.")
    json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
    print(name, {k: (round(v["frac_of_peak"], 4) if isinstance(v, dict) and "frac_of_peak" in v else None)
                 for k, v in summary.get("kernels", {}).items()})


def off_diag_flops(F):
    w = [min(256, F - 256 * i) for i in range((F + 255) // 256)]
    off = 2.0 * sum(w[i] * w[j] for i in range(len(w)) for j in range(i + 1, len(w)))
    return off, F * (F + 1.0) - off


# ---- headline: N = 10M, 5 launches of 2M rows per pass ----
b = bench_record("headline_kt")
if b:
    st, tr = kernel_stats("headline_kt"), kernel_trace("headline_kt")
    F, d = 4096, 32
    rows, lp = b["roofline"]["rows_per_step"], b["roofline"]["launches_per_step"]
    off, dg = off_diag_flops(F)
    ks = {"rr_syrk_f32_kernel": by_totals(st, tr, "rr_syrk_f32_kernel(", off, rows, lp, PEAK["f32"], "flop", "off-diagonal 256x256 tiles of Phi^T Phi"),
          "rr_syrk_f32_diag16_kernel": by_totals(st, tr, "rr_syrk_f32_diag16_kernel", dg, rows, lp, PEAK["f32"], "flop", "diagonal tiles"),
          "rr_rff_features_mfma_kernel": by_totals(st, tr, "rr_rff_features_mfma_kernel", 4.0 * d + 4.0 + 4.0 * F, rows, lp, PEAK["hbm"], "byte",
                                                   "X Ws on MFMA + sin / cos -> P (HBM write), Phi^T y")}
    put("headline", "headline_kt", {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps %d --warmup %d --configs none (tools/prof.sh headline)" % (b["steps"], b["warmup"]),
                                    "bench_line": {k: b[k] for k in ("value", "ms_per_step", "roofline")}, "kernels": ks, "all_kernels": st})

# ---- one _elbo evaluation at config 2's shape (f32, N = 1M; f64, N = 200k) ----
for tag, name, key, dt in (("elbo_kt", "elbo", "C2_elbo_eval", "f32"), ("elbo64_kt", "elbo_f64", "C2f64_elbo_eval_n200k", "f64")):
    b = bench_record(tag)
    if not b or "error" in b.get("configs", {}).get(key, {"error": 1}):
        continue
    cfg, st, tr = b["configs"][key], kernel_stats(tag), kernel_trace(tag)
    N, F, d = cfg["rows"], 4096, 32
    off, dg = off_diag_flops(F)
    p2 = 2.0 * F * F + 2.0 * d * F
    ks = {}
    if dt == "f32":
        p2chunks = -(-N // 524288)
        ks["rr_syrk_f32_kernel"] = by_totals(st, tr, "rr_syrk_f32_kernel(", off, N, 1, PEAK["f32"], "flop", "statistics pass, off-diagonal tiles")
        ks["rr_syrk_f32_diag16_kernel"] = by_totals(st, tr, "rr_syrk_f32_diag16_kernel", dg, N, 1, PEAK["f32"], "flop", "statistics pass, diagonal tiles")
        ks["rr_gemm_gradt_f32_kernel"] = by_totals(st, tr, "rr_gemm_gradt_f32_kernel", p2, N, p2chunks, PEAK["f32"], "flop",
                                                   "second pass: U = Phi C and T = X^T ((Err m^T - U) o dPhi-pattern) in one kernel, %d EQUAL "
                                                   "row chunks per pass: 2 F^2 + 2 d F flop per row" % p2chunks)
    else:
        # f64: the Gram and the GEMM kernels also serve the posterior (small grids): the passes' launches by grid size
        ks["rr_syrk_f64_kernel"] = by_totals(st, tr, "rr_syrk_f64_kernel(", F * F - 128.0 * F, N, 1, PEAK["f64"], "flop",
                                             "statistics pass (off-diagonal 128x128 tiles)", min_grid="max")
        ks["rr_gemm_tn_f64_kernel"] = by_totals(st, tr, "rr_gemm_tn_f64_kernel", 2.0 * F * F, N, 1, PEAK["f64"], "flop",
                                                "second pass: U = Phi C", min_grid="max")
    npost = max(sum(v["calls"] for k, v in st.items() if k.startswith("rr_posterior_rows_kernel")), 1)
    post = {k.split("(")[0]: v["total_ms"] / npost for k, v in st.items()
            if k.replace("void ", "").startswith(("rr_gemm_tn_f64", "rr_chol_diag", "rr_syrk_f64", "rr_posterior_rows", "rr_assemble_ic", "rr_extract"))}
    put(name, tag, {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --rows 1000000 --steps 1 --warmup 0 --configs %s (tools/prof.sh)" % key.lower(),
                    "config": cfg, "kernels": {k: v for k, v in ks.items() if v}, "posterior_kernels_ms_per_call": post, "all_kernels": st})

# ---- config 3's one-GPU share: concatenated Gram + second pass ----
b = bench_record("c3_kt")
key = "C3_matern52_linear_concat_one_gpu_share"
if b and "error" not in b.get("configs", {}).get(key, {"error": 1}):
    cfg, st, tr = b["configs"][key], kernel_stats("c3_kt"), kernel_trace("c3_kt")
    N, Ft, n3, d3 = cfg["rows"], 8257, 4096, 64
    ch = len(cfg["rows_per_launch"])
    off, dg = off_diag_flops(8192)  # the tiles among the first 32 column blocks; the ragged 33rd block has its own kernel
    ks = {"rr_syrk_f32_kernel": by_totals(st, tr, "rr_syrk_f32_kernel(", off, N, ch, PEAK["f32"], "flop", "Gram pass, off-diagonal tiles of the 32 full column blocks", min_grid="mode"),
          "rr_gemm_gradt_f32_kernel": by_totals(st, tr, "rr_gemm_gradt_f32_kernel", 2.0 * Ft * 2 * n3 + 2.0 * d3 * 2 * n3, N, ch, PEAK["f32"], "flop",
                                                "second pass: the random Fourier child's 32 column tiles of U = Phi C (K = all 8257 features) contracted "
                                                "in registers with both 32-column blocks of X; the linear child's columns are not computed")}
    put("c3", "c3_kt", {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --rows 1000000 --steps 1 --warmup 0 --configs c3 (tools/prof.sh c3)",
                        "config": cfg, "kernels": {k: v for k, v in ks.items() if v}, "all_kernels": st})

# ---- FastFoodRBF _elbo at F = 16384 ----
b = bench_record("ffelbo_kt")
key = "C4elbo_fastfood_f16384"
if b and "error" not in b.get("configs", {}).get(key, {"error": 1}):
    cfg, st, tr = b["configs"][key], kernel_stats("ffelbo_kt"), kernel_trace("ffelbo_kt")
    N, F, d = cfg["rows"], 16384, 128
    ch = len(cfg.get("rows_per_launch") or [0] * (-(-N // 131072)))
    off, dg = off_diag_flops(F)
    ks = {"rr_fastfood16_kernel": by_totals(st, tr, "rr_fastfood16_kernel", 4.0 * d + 4.0 * F, N, ch, PEAK["hbm"], "byte",
                                            "the chain writes Phi into the feature matrix (both passes: launches_per_pass counts one of them; the one extra "
                                            "launch makes the dense equivalent W = _makeVX(I_d) for the gradient)", min_grid="mode"),
          "rr_syrk_f32_kernel": by_totals(st, tr, "rr_syrk_f32_kernel(", off, N, ch, PEAK["f32"], "flop", "statistics pass, off-diagonal tiles at F = 16384", min_grid="mode"),
          "rr_syrk_f32_diag16_kernel": by_totals(st, tr, "rr_syrk_f32_diag16_kernel", dg, N, ch, PEAK["f32"], "flop", "statistics pass, diagonal tiles", min_grid="mode"),
          "rr_gemm_gradt_f32_kernel": by_totals(st, tr, "rr_gemm_gradt_f32_kernel", 2.0 * F * F + 2.0 * d * F, N, ch, PEAK["f32"], "flop",
                                                "second pass: U = Phi C contracted in registers with the four 32-column blocks of X")}
    npost = max(sum(v["calls"] for k, v in st.items() if k.startswith("rr_posterior_rows_kernel")), 1)
    post = {k.split("(")[0]: v["total_ms"] / npost for k, v in st.items()
            if k.replace("void ", "").startswith(("rr_gemm_tn_f64", "rr_chol_diag", "rr_syrk_f64", "rr_posterior_rows", "rr_assemble_ic", "rr_extract"))}
    put("ff_elbo", "ffelbo_kt", {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --rows 1000000 --steps 1 --warmup 0 --configs c4elbo (tools/prof.sh ffelbo)",
                                 "config": cfg, "kernels": {k: v for k, v in ks.items() if v}, "posterior_kernels_ms_per_call": post,
                                 "posterior_note": "kernel durations summed over the call's three concurrent streams (they exceed its wall-clock)",
                                 "all_kernels": st})

# ---- the rest: stats + the bench entry ----
for tag, name, keys in (("posdef_kt", "posdef", ("posterior_F4096", "posterior_F8257", "posterior_F16384")),
                        ("predict_kt", "predict", ("predict_moments_n300k",)), ("laplace_kt", "laplace", ("C2laplace_f64phase_n1m",)),
                        ("c4_kt", "c4_fastfood", ("C4_fastfood_f16384",)), ("c5_kt", "c5_glm", ("C5_glm_poisson_svi_step",)),
                        ("c1_kt", "c1_latency", ("C1_elbo_latency",)), ("c4gm_kt", "c4gm", ("C4gm_fastfoodgm_f16384",)),
                        ("c5small_kt", "glm_small_batch", ("C5small_glm_default_fit",)),
                        ("sp_kt", "single_process", ())):
    b = bench_record(tag)
    if not b:
        continue
    st, tr = kernel_stats(tag), kernel_trace(tag)
    cfgs = {k: b.get("configs", {}).get(k) for k in keys}
    ks = {}
    if name == "c4_fastfood" and cfgs[keys[0]] and "error" not in cfgs[keys[0]]:
        c = cfgs[keys[0]]
        ks["rr_fastfood16_kernel"] = by_totals(st, tr, "rr_fastfood16_kernel", c["roofline"]["bytes_per_row"], c["rows"], 16, PEAK["hbm"], "byte",
                                               "FastFood chain -> Phi (HBM write), 16 launches of 262 144 rows per pass")
    if name == "c4gm" and cfgs[keys[0]] and "error" not in cfgs[keys[0]]:
        c = cfgs[keys[0]]
        ks["rr_fastfood16_kernel<..., GM>"] = by_totals(st, tr, "rr_fastfood16_kernel", c["roofline"]["bytes_per_row"], c["rows"], 8, PEAK["hbm"], "byte",
                                                        "FastFoodGM: the chain's mixture mode -> four trig blocks (HBM write), 8 launches of 262 144 rows "
                                                        "per pass, three passes (warm-up + 2 timed) -- the first 24 launches of the kernel's most frequent grid; the "
                                                        "_elbo part that follows launches it on 131 072 rows", min_grid="mode", first_n=24)
    if name == "single_process" and b.get("config", {}).get("single_process"):
        off, dg = off_diag_flops(4096)
        rows, lp, n = b["roofline"]["rows_per_step"], b["roofline"]["launches_per_step"], b["n_gpus"]
        ks["rr_syrk_f32_kernel"] = by_totals(st, tr, "rr_syrk_f32_kernel(", off, rows * n, lp * n, PEAK["f32"], "flop",
                                             "headline step of bench.py --single-process: every member's launches (members share the one GPU of "
                                             "the box, so their kernels time-share it: fraction of ONE GPU's peak)", min_grid="max")
        cfgs = {"line": {k: b.get(k) for k in ("value", "ms_per_step", "n_gpus", "exchange", "per_rank")}, "configs": b.get("configs")}
    if name == "c5_glm" and cfgs[keys[0]] and "error" not in cfgs[keys[0]]:
        KL, Fq, Mq = 500, 2048, cfgs[keys[0]]["rows_per_step"]
        # rr_gemm_tn_f32_kernel also serves the posterior-free statistics part of the bench's headline (other grids): the step's
        # launches are the kernel's most frequent grid
        for kn, sel, what in (("rr_gemm_lik_f32_kernel", None, "fs = Phi ws^T, likelihood terms as the epilogue, dfs stored in both layouts"),
                              ("rr_gemm_tn_f32_kernel", "mode", "Ed = dfs Phi: 16 output tiles, K = 65 536 minibatch rows split 16 ways (f32 atomics)"),
                              ("rr_gemm_gradt_f32_kernel", None, "EdPhi = dfs^T ws contracted with P and X in registers (never stored)")):
            ks[kn] = by_totals(st, tr, kn, 2.0 * KL * Fq, Mq, 1, PEAK["f32"], "flop",
                               what + "; 2 K L F algorithmic flop per minibatch row and launch (K L = 500 sample columns, 512 issued)", min_grid=sel)
    if name == "predict" and cfgs[keys[0]] and "error" not in cfgs[keys[0]]:
        Fq = 4096
        ks["rr_gemm_pair_f32_kernel"] = by_totals(st, tr, "rr_gemm_pair_f32_kernel", 1.0 * Fq * Fq, cfgs[keys[0]]["rows"], 1, PEAK["f32"], "flop",
                                                  "Phi B with the upper-triangular factor B, column tiles paired into equal-cost workgroups, row sums of "
                                                  "squares in the epilogue: F^2 algorithmic flop per row (1.016 F^2 issued: in a diagonal block a wave skips the "
                                                  "k-blocks its column quarter of B is zero in; 1.0625 F^2 before that); the full-size calls", min_grid="max")
    put(name, tag, {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --rows 1000000 --steps 1 --warmup 0 --configs %s (tools/prof.sh)" % ",".join(k.lower() for k in keys),
                    "configs": cfgs, "kernels": {k: v for k, v in ks.items() if v}, "all_kernels": st})


# ---- counter passes: matrix-pipe busy cycles and clock per kernel (tools/prof.sh sq) ----
def sq_summary(tag, prefixes, min_ms):
    import collections
    pth = os.path.join(SRC, tag, "p_counter_collection.csv")
    if not os.path.exists(pth):
        return None
    d, t, nm = collections.defaultdict(lambda: collections.defaultdict(float)), {}, {}
    for r in csv.DictReader(open(pth)):
        n = r["Kernel_Name"].replace("void ", "")
        if n.startswith(prefixes):
            d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            t[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            nm[r["Dispatch_Id"]] = n.split("(")[0]
    out = {}
    for k in d:
        if t[k] < min_ms:
            continue
        ghz = d[k]["GRBM_GUI_ACTIVE"] / 8.0 / (t[k] * 1e6)
        busy = d[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4.0 * d[k]["SQ_BUSY_CU_CYCLES"], 1.0)  # four SIMDs (matrix pipes) per CU
        e = out.setdefault(nm[k], {"launches": 0, "ms": [], "clock_GHz": [], "mfma_busy_frac": []})
        e["launches"] += 1
        e["ms"].append(t[k])
        e["clock_GHz"].append(ghz)
        e["mfma_busy_frac"].append(busy)
    for e in out.values():
        # (the clock of ~1 ms launches follows the chip's power state: a burst after an idle gap runs at ~2.4 GHz, a sustained
        # queue of MFMA-bound launches at ~2.1 GHz -- the range is part of the evidence)
        e["clock_GHz_min_max"] = [min(e["clock_GHz"]), max(e["clock_GHz"])]
        e["ms_min_max"] = [min(e["ms"]), max(e["ms"])]
        for key in ("ms", "clock_GHz", "mfma_busy_frac"):
            e[key] = sum(e[key]) / len(e[key])
        e["ceiling_frac_of_nominal_peak"] = e["mfma_busy_frac"] * e["clock_GHz"] / 2.4
    return out or None


for tag, name in (("headline_sq", "headline"), ("elbo_sq", "elbo"), ("c3_sq", "c3"), ("predict_sq", "predict"), ("c5_sq", "c5_glm")):
    if name == "c5_glm":  # the GLM step's three products: ~1 ms launches, K = 2048 or a 1/16 K-split of 65 536
        sq = sq_summary(tag, ("rr_gemm_lik_f32_kernel", "rr_gemm_gradt_f32_kernel", "rr_gemm_tn_f32_kernel"), 0.5)
    else:
        sq = sq_summary(tag, ("rr_syrk_f32_kernel", "rr_syrk_f32_diag16_kernel", "rr_gemm_gradt_f32_kernel", "rr_gemm_tn_f32_kernel", "rr_gemm_pair_f32_kernel"), 5.0)
    if sq:
        dst = os.path.join(ROOT, "profiles", "%s_%s" % (ROUND, name))
        os.makedirs(dst, exist_ok=True)
        json.dump({"command": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES "
                              "SQ_INSTS_VALU_MFMA_MOPS_F32 -- python bench.py ... (tools/prof.sh sq)",
                   "clock": "GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md)",
                   "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)",
                   "ceiling_frac_of_nominal_peak": "mfma_busy_frac x clock / 2.4 GHz: what the roofline fraction of a kernel that issues only useful MFMAs can be",
                   "kernels": sq},
                  open(os.path.join(dst, "pmc_sq.json"), "w"), indent=1)
        print(name, "pmc_sq", {k: (round(v["mfma_busy_frac"], 4), round(v["clock_GHz"], 3)) for k, v in sq.items()})


# ---- L2 -> fabric traffic of the headline kernels (tools/prof.sh hbm): profiles/traffic.json, quoted by bench.py ----
def _per_dispatch(tag, prefix, ctr):
    import collections
    pth = os.path.join(SRC, tag, "p_counter_collection.csv")
    vals = collections.OrderedDict()
    if os.path.exists(pth):
        for r in csv.DictReader(open(pth)):
            if r["Kernel_Name"].replace("void ", "").startswith(prefix) and r["Counter_Name"] == ctr:
                vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return list(vals.values())


fv, wv = _per_dispatch("headline_fetch", "rr_syrk_f32_kernel(", "FETCH_SIZE"), _per_dispatch("headline_write", "rr_syrk_f32_kernel(", "WRITE_SIZE")
bf = bench_record("headline_fetch")
if fv and wv and bf:
    rows = bf["roofline"]["rows_per_step"] // max(bf["roofline"]["launches_per_step"], 1)
    # MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE and WRITE_SIZE are in KiB; gfx950 counts wide coalesced reads at
    # half their size (x 2 on the fetch side); separate --pmc passes; per launch
    fetch, write = sum(fv) / len(fv) * 1024 * 2, sum(wv) / len(wv) * 1024
    tr = {"kernel": "rr_syrk_f32_kernel", "rows_per_launch": rows, "fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write,
          "fetch_bytes_of_each_launch": [x * 2048 for x in fv], "bytes_per_row": (fetch + write) / rows,
          "feature_matrix_bytes_per_launch": rows * 4096 * 4.0,
          "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, the round's binaries, the default command's own %d-row launches, "
                  "average of a pass' launches); FETCH_SIZE*1024*2 (gfx950 half-count correction of wide coalesced reads, "
                  "MI355X_MICROARCH.md) + WRITE_SIZE*1024; L2->fabric side: requests the Infinity Cache serves are counted too "
                  "(profiles/r02_mall has the probe); profiles/%s_headline" % (rows, ROUND)}
    json.dump(tr, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    extra = {}
    for pref in ("rr_syrk_f32_diag16_kernel", "rr_rff_features_mfma_kernel"):
        f2, w2 = _per_dispatch("headline_fetch", pref, "FETCH_SIZE"), _per_dispatch("headline_write", pref, "WRITE_SIZE")
        if f2 and w2:
            extra[pref] = {"fetch_bytes_per_launch": sum(f2) / len(f2) * 2048, "write_bytes_per_launch": sum(w2) / len(w2) * 1024}
    p2 = os.path.join(ROOT, "profiles", "%s_headline" % ROUND, "summary.json")
    if os.path.exists(p2):
        sm = json.load(open(p2))
        sm["hbm_side_traffic"] = {"rr_syrk_f32_kernel": tr, **extra}
        json.dump(sm, open(p2, "w"), indent=1)
    for tag, name in (("headline_fetch", "pmc_fetch.csv"), ("headline_write", "pmc_write.csv")):
        src = os.path.join(SRC, tag, "p_counter_collection.csv")
        if os.path.exists(src) and os.path.getsize(src) < 4 << 20:
            shutil.copy(src, os.path.join(ROOT, "profiles", "%s_headline" % ROUND, name))
    print("traffic.json", {k: tr[k] for k in ("rows_per_launch", "fetch_bytes", "write_bytes", "bytes_per_row")})
