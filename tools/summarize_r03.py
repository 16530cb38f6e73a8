#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a tools/prof_r03.sh run (gpurun_out/prof_r03) into profiles/r03_<name>/ and derive a
summary.json per profile: every kernel's calls / average / total from rocprofv3's own stats, and -- for the kernels that
carry the flops or bytes of the profiled configuration -- rows per launch, algorithmic work per launch, the achieved rate
from rocprofv3's average duration of the FULL-SIZE launches and the fraction of the peak that bounds them.

    python tools/summarize_r03.py
"""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r03")
PEAK = {"f32": 157.3e12, "f64": 78.6e12, "hbm": 8.0e12}


def bench_line(tag):
    p = os.path.join(SRC, tag + ".json")
    if not os.path.exists(p):
        return None
    lines = [l for l in open(p) if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


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
    p = os.path.join(SRC, tag, "kt_kernel_trace.csv")
    out = []
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            out.append((r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                        int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return out


def full_size(durs, frac=0.9):
    """(average, count) over the launches within `frac` of the longest-half median: the full-size ones."""
    durs = sorted(durs)
    if not durs:
        return None, 0
    ref = durs[len(durs) // 2:]
    ref = ref[len(ref) // 2]
    top = [d for d in durs if d > frac * ref and d < ref / frac]
    return sum(top) / len(top), len(top)


def put(name, tag, summary, extra=()):
    dst = os.path.join(ROOT, "profiles", name)
    os.makedirs(dst, exist_ok=True)
    files = [(os.path.join(SRC, tag, "kt_kernel_stats.csv"), "kernel_stats.csv"),
             (os.path.join(SRC, tag + ".json"), "bench_under_rocprof.json")] + list(extra)
    for src, dname in files:
        if os.path.exists(src):
            shutil.copy(src, os.path.join(dst, dname))
    json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
    print(name, json.dumps(summary)[:400])


def annotate(trace, prefix, work, peak, unit, rows=None, what="", select="full", min_ms=0.0):
    """select: "full" = the full-size launches (see full_size); "all" = every launch (work = the AVERAGE launch's);
    "top3" = the three longest.  min_ms drops short launches of the same kernel that belong to another stage."""
    durs = [d for n, d, _, _ in trace if n.replace("void ", "").startswith(prefix) and d >= min_ms]
    if select == "all":
        avg, nfull = (sum(durs) / len(durs), len(durs)) if durs else (None, 0)
    elif select == "top3":
        top = sorted(durs)[-3:]
        avg, nfull = (sum(top) / len(top), len(top)) if top else (None, 0)
    else:
        avg, nfull = full_size(durs)
    if avg is None:
        return None
    rate = work / (avg * 1e-3)
    return {"full_size_launches": nfull, "all_launches": len(durs), "avg_ms_full_size": avg, "rows_per_launch": rows,
            "launches_selected": select, "algorithmic_work_per_launch": work, "unit": unit, "achieved": rate / (1e12 if unit == "flop" else 1e9),
            "achieved_unit": "TFLOP/s" if unit == "flop" else "GB/s", "frac_of_peak": rate / peak, "what": what}


F, d, n = 4096, 32, 2048

# ---- headline: N = 10M, 5 launches of 2M rows per pass ----
for tag, name in (("headline_kt", "r03_headline"), ("overlap_off_kt", "r03_overlap"), ("det_kt", "r03_deterministic")):
    b = bench_line(tag)
    if not b:
        continue
    st, tr = kernel_stats(tag), kernel_trace(tag)
    rows = b["roofline"]["rows_per_step"] // max(b["roofline"]["launches_per_step"], 1)
    off = b["roofline"]["flops_per_row"]
    # a 10M-row pass is 4 launches of 2 097 152 rows (32 GiB of P) and one of 1 611 392: ALL launches are averaged, on the
    # average launch's rows (= rows per step / launches per step, as the bench line's HIP events do)
    dk = "rr_syrk_f32_diag16_kernel" if any("rr_syrk_f32_diag16_kernel" in nm for nm, _, _, _ in tr) else "rr_syrk_f32_diag_kernel("
    ks = {"rr_syrk_f32_kernel": annotate(tr, "rr_syrk_f32_kernel(", off * rows, PEAK["f32"], "flop", rows, "off-diagonal 256x256 tiles of Phi^T Phi", "all"),
          dk.rstrip("("): annotate(tr, dk, (F * (F + 1.0) - off) * rows, PEAK["f32"], "flop", rows, "diagonal tiles", "all"),
          "rr_rff_features_mfma_kernel": annotate(tr, "rr_rff_features_mfma_kernel", rows * (4.0 * d + 4.0 + 4.0 * F), PEAK["hbm"], "byte", rows,
                                                  "X Ws on MFMA + sin / cos -> P (HBM write), Phi^T y", "all")}
    summary = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps %d --warmup %d --configs none (tools/prof_r03.sh)" % (b["steps"], b["warmup"]),
               "bench_line": {k: b[k] for k in ("value", "ms_per_step", "roofline")}, "kernels": ks, "all_kernels": st}
    if tag == "headline_kt" and tr:
        # how much of the feature kernels' time ran UNDER a SYRK kernel (second stream): overlap of the intervals
        feats = [(s, e) for nme, _, s, e in tr if "rr_rff_features_mfma_kernel" in nme]
        syrk = [(s, e) for nme, _, s, e in tr if "rr_syrk_f32" in nme]
        hidden = sum(max(0, min(e, e2) - max(s, s2)) for s, e in feats for s2, e2 in syrk) / 1e6
        summary["feature_kernel_ms_total"] = sum(e - s for s, e in feats) / 1e6
        summary["feature_kernel_ms_under_a_syrk_kernel"] = hidden
    extra = []
    if tag == "overlap_off_kt":
        ab = {}
        for arm in ("off", "on"):
            vals = []
            for rep in (1, 2):
                bl = bench_line("overlap_%s_%d" % (arm, rep))
                if bl:
                    vals.append({"value": bl["value"], "ms_per_step": bl["ms_per_step"], "whole_path_frac": bl["roofline"]["whole_path_frac"],
                                 "syrk_ms": bl["roofline"]["kernel_ms_per_step"], "other": bl["roofline"]["other_kernels_ms_per_step"]})
            ab["RR_GRAM_OVERLAP=%d" % (arm == "on")] = vals
        summary["ab_unprofiled_bench_lines"] = ab
    put(name, tag, summary, extra)

# ---- one _elbo evaluation (f32, N = 1M; f64, N = 200k) ----
for tag, name, key, peak, es in (("elbo_kt", "r03_elbo", "C2_elbo_eval", PEAK["f32"], 4), ("elbo64_kt", "r03_elbo_f64", "C2f64_elbo_eval_n200k", PEAK["f64"], 8)):
    b = bench_line(tag)
    if not b or key not in b.get("configs", {}) or "error" in b["configs"][key]:
        continue
    cfg, st, tr = b["configs"][key], kernel_stats(tag), kernel_trace(tag)
    N = cfg["rows"]
    Fp = F
    p2rows = min(N, ((24 << 30) // (12 * Fp) + 255) // 256 * 256) if es == 4 else None
    ks = {}
    if es == 4:
        ks["rr_syrk_f32_kernel"] = annotate(tr, "rr_syrk_f32_kernel(", (F * (F + 1.0) - 16 * 256 * 257.0) * N, peak, "flop", N, "statistics pass")
        ks["rr_gemm_tn_f32_kernel"] = annotate(tr, "rr_gemm_tn_f32_kernel", 2.0 * F * F * p2rows, peak, "flop", p2rows,
                                               "second pass: U = Phi C, one launch per 524 288-row chunk (the launches above 50 ms; "
                                               "the 256-row parity slice runs the same kernel)", "full", 50.0)
        ks["rr_grad_t_kernel"] = annotate(tr, "rr_grad_t_kernel", (4.0 * d + 8.0 * F) * p2rows, PEAK["hbm"], "byte", p2rows,
                                          "second pass: T = X^T A, reads P and U once (HBM read bound)", "full", 1.0)
        # since the third session of round 3 the two kernels above are ONE: U = Phi C contracted in registers
        ks["rr_gemm_gradt_f32_kernel"] = annotate(tr, "rr_gemm_gradt_f32_kernel", (2.0 * F * F + 2.0 * d * F) * p2rows, peak, "flop", p2rows,
                                                  "second pass: U = Phi C and T = X^T ((Err m^T - U) o dPhi-pattern) in one kernel, one launch "
                                                  "per 524 288-row chunk: 2 F^2 + 2 d F flop per row", "full", 50.0)
        ks = {k: v for k, v in ks.items() if v is not None}
        # the posterior's kernels: totals per rr_posterior_dev call
        npost = max(st.get("rr_posterior_rows_kernel(double const*, double const*, double const*, double, long, double*, double*, double*, long)", {}).get("calls", 0), 1)
        ks["posterior_kernels_ms_per_call"] = {k.split("(")[0]: v["total_ms"] / npost for k, v in st.items()
                                               if k.startswith(("rr_gemm_tn_f64", "rr_chol_diag", "rr_syrk_f64", "rr_posterior_rows", "rr_trsm", "rr_assemble_ic", "rr_extract"))}
    else:
        ks["rr_syrk_f64_kernel"] = annotate(tr, "rr_syrk_f64_kernel(", (F * F - 128.0 * F) * N, peak, "flop", N, "statistics pass (off-diagonal tiles)")
        ks["rr_gemm_tn_f64_kernel"] = annotate(tr, "rr_gemm_tn_f64_kernel", 2.0 * F * F * N, peak, "flop", N,
                                               "second pass: U = Phi C (the launches above 10 ms; the posterior's trailing updates use the same kernel)", "full", 10.0)
        ks["rr_syrk_f64_kernel"] = annotate(tr, "rr_syrk_f64_kernel(", (F * F - 128.0 * F) * N, peak, "flop", N, "statistics pass (off-diagonal tiles)", "full", 10.0)
    put(name, tag, {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --rows 1000000 --steps 1 --warmup 0 --configs %s" % key.lower(),
                    "config": cfg, "kernels": ks, "all_kernels": st})

# ---- the rest: stats + the bench entry ----
for tag, name, keys in (("posdef_kt", "r03_posdef", ("posterior_F4096", "posterior_F8257")), ("predict_kt", "r03_predict", ("predict_moments_n300k",)),
                        ("laplace_kt", "r03_laplace", ("C2laplace_f64phase_n1m",)), ("c4_kt", "r03_c4_fastfood", ("C4_fastfood_f16384",)),
                        ("c5_kt", "r03_c5_glm", ("C5_glm_poisson_svi_step",)), ("c3_kt", "r03_c3", ("C3_matern52_linear_concat_one_gpu_share",))):
    b = bench_line(tag)
    if not b:
        continue
    st, tr = kernel_stats(tag), kernel_trace(tag)
    cfgs = {k: b.get("configs", {}).get(k) for k in keys}
    ks = {}
    if name == "r03_laplace" and cfgs[keys[0]]:
        N = cfgs[keys[0]]["rows"]
        ks["rr_rff_features_mfma64_kernel"] = annotate(tr, "rr_rff_features_mfma64_kernel", N * (8.0 * d + 8.0 + 4.0 * F), PEAK["hbm"], "byte", N,
                                                       "float64 X in, float64 phases on the f64 MFMA, float32 sin / cos, float32 P out")
        ks["rr_syrk_f32_kernel"] = annotate(tr, "rr_syrk_f32_kernel(", (F * (F + 1.0) - 16 * 256 * 257.0) * N, PEAK["f32"], "flop", N, "same SYRK as RandomRBF")
    if name == "r03_predict" and cfgs[keys[0]]:
        Np = (cfgs[keys[0]]["rows"] + 255) // 256 * 256
        ks["rr_gemm_tn_f32_kernel"] = annotate(tr, "rr_gemm_tn_f32_kernel", 1.0 * F * F * Np, PEAK["f32"], "flop", Np,
                                               "Phi B with the upper-triangular factor B, ONE launch per predict_moments call: F^2 flop per "
                                               "row (half of 2 F^2; the three longest launches = the three timed calls)", "top3")
    if name == "r03_c5_glm" and cfgs[keys[0]]:
        M5, F5, KL5, d5 = 65536, 2048, 500, 32
        gf = 2.0 * M5 * F5 * KL5
        for kn, what, work in (("rr_gemm_lik_f32_kernel", "fs = Phi ws^T with the likelihood terms as its epilogue (dfs stored in both layouts)", gf),
                               ("rr_gemm_tn_f32_kernel", "Ed = dfs Phi (K-split over the minibatch rows, f32 atomics)", gf),
                               ("rr_gemm_gradt_f32_kernel", "EdPhi = dfs^T ws contracted with P and X in registers (never stored)", gf + 2.0 * d5 * F5 * M5)):
            a = annotate(tr, kn, work, PEAK["f32"], "flop", M5, what, "all", 0.5)
            if a:
                ks[kn] = a
    if name == "r03_c3" and cfgs[keys[0]]:
        Ft, n3, d3 = 8257, 4096, 64
        rows3 = cfgs[keys[0]].get("rows_per_launch", 254200)
        a = annotate(tr, "rr_gemm_gradt_f32_kernel", (2.0 * Ft * 2 * n3 + 2.0 * d3 * 2 * n3) * rows3, PEAK["f32"], "flop", rows3,
                     "second pass: the random Fourier child's 32 column tiles of U = Phi C (K = all 8257 features) contracted in "
                     "registers; the linear child's columns are not computed", "full", 50.0)
        if a:
            ks["rr_gemm_gradt_f32_kernel"] = a
    if name == "r03_c4_fastfood" and cfgs[keys[0]]:
        r = cfgs[keys[0]]["roofline"]
        ks["rr_fastfood16_kernel"] = annotate(tr, "rr_fastfood16_kernel", r["bytes_per_row"] * r["rows_per_launch"], PEAK["hbm"], "byte", r["rows_per_launch"],
                                              "FastFood chain -> Phi (HBM write)")
    put(name, tag, {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --rows 1000000 --steps 1 --warmup 0 --configs %s" % ",".join(k.lower() for k in keys),
                    "configs": cfgs, "kernels": ks, "all_kernels": st})


# ---- the GEMM kernel's clock and cycle efficiency in the GLM step vs the SLM second pass (PMC pass: tools/prof_r03.sh gemmclk) ----
def gemm_clock(tag, min_us, ideal_cycles):
    import collections
    pth = os.path.join(SRC, tag, "p_counter_collection.csv")
    if not os.path.exists(pth):
        return None
    d, t = collections.defaultdict(lambda: collections.defaultdict(float)), {}
    for r in csv.DictReader(open(pth)):
        if r["Kernel_Name"].startswith("rr_gemm_tn_f32"):
            d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            t[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    sel = [k for k in d if t[k] >= min_us and (ideal_cycles is None or t[k] < 3000)] if ideal_cycles else [k for k in d if t[k] >= min_us]
    if not sel:
        return None
    ghz = [d[k]["GRBM_GUI_ACTIVE"] / 8.0 / (t[k] * 1e3) for k in sel]
    cyc = [d[k]["GRBM_GUI_ACTIVE"] / 8.0 for k in sel]
    return {"launches": len(sel), "avg_us": sum(t[k] for k in sel) / len(sel), "clock_GHz_avg": sum(ghz) / len(ghz),
            "clock_GHz_min_max": [min(ghz), max(ghz)], "cycles_per_launch_avg": sum(cyc) / len(cyc),
            "note": "clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS)"}


g5 = gemm_clock("c5_sq", 800.0, 1)
ge = gemm_clock("elbo_sq", 120000.0, None)  # the 524 288-row launches (the 475 712-row remainder launches take 112 ms)
if g5 and ge:
    # ideal MFMA cycles per launch: k-blocks x 16 k-steps x 16 MFMAs x 64 cycles per SIMD, rounds of 256 workgroups
    g5["ideal_cycles"] = 2 * 64 * 16384.0          # 512 tiles = 2 rounds, K = 2048 = 64 k-blocks
    g5["cycle_efficiency"] = g5["ideal_cycles"] / g5["cycles_per_launch_avg"]
    rounds = ge["avg_us"]  # placeholder to keep the structure simple below
    ge["ideal_cycles"] = 128 * 128 * 16384.0       # 32 768 tiles = 128 rounds, K = 4096 = 128 k-blocks (524 288-row launches)
    ge["cycle_efficiency"] = ge["ideal_cycles"] / ge["cycles_per_launch_avg"]
    out = {"what": "rr_gemm_tn_f32_kernel in one SVI step of config 5 (three launches per step) against the same kernel in the SLM "
                   "second pass: the GLM launches run at a LOWER CLOCK (the chip's DVFS under a bursty 4 ms step) and lose a few "
                   "more cycles to their two-round, short-K shape",
           "glm_step_gemms": g5, "slm_second_pass_gemm": ge,
           "time_ratio_explained": {"clock": ge["clock_GHz_avg"] / g5["clock_GHz_avg"], "cycles": ge["cycle_efficiency"] / g5["cycle_efficiency"]}}
    dst = os.path.join(ROOT, "profiles", "r03_c5_glm")
    os.makedirs(dst, exist_ok=True)
    json.dump(out, open(os.path.join(dst, "gemm_clock.json"), "w"), indent=1)
    for tag, name in (("c5_sq", "pmc_sq_c5.csv"), ("elbo_sq", "pmc_sq_elbo.csv")):
        src = os.path.join(SRC, tag, "p_counter_collection.csv")
        if os.path.getsize(src) < 3 << 20:
            shutil.copy(src, os.path.join(dst, name))
    print("r03_c5_glm/gemm_clock.json", json.dumps(out)[:600])


# ---- L2 -> fabric traffic of the headline kernels (tools/prof_r03.sh hbm): profiles/traffic.json, quoted by bench.py ----
def _per_dispatch(tag, prefix, ctr):
    import collections
    pth = os.path.join(SRC, tag, "p_counter_collection.csv")
    vals = collections.defaultdict(float)
    if os.path.exists(pth):
        for r in csv.DictReader(open(pth)):
            if r["Kernel_Name"].replace("void ", "").startswith(prefix) and r["Counter_Name"] == ctr:
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return list(vals.values())


fv, wv = _per_dispatch("headline_fetch", "rr_syrk_f32_kernel(", "FETCH_SIZE"), _per_dispatch("headline_write", "rr_syrk_f32_kernel(", "WRITE_SIZE")
fv10 = _per_dispatch("headline10m_fetch", "rr_syrk_f32_kernel(", "FETCH_SIZE")  # the default command's own launches (5 x 2M rows)
if fv and wv:
    fv = [x for x in fv if x > 0.5 * max(fv)]
    wv = [x for x in wv if x > 0.5 * max(wv)] or wv
    fetch, write = sum(fv) / len(fv) * 1024 * 2, sum(wv) / len(wv) * 1024
    rows = 2097152
    pow2 = {"rows_per_launch": rows, "fetch_bytes": fetch,
            "note": "a 2^21-row launch (RR_GRAM_CHUNK_ROWS=2097152, the chunking before round 3's equal chunks): its K-splits sit "
                    "2^15 rows = 512 MiB apart and alias in the L2"}
    if fv10:
        write = write * 2000000.0 / rows   # the flush is per output tile and split: scales with the splits, i.e. the rows
        fetch, rows = sum(fv10) / len(fv10) * 1024 * 2, 2000000
    tr = {"kernel": "rr_syrk_f32_kernel", "rows_per_launch": rows, "fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write,
          "fetch_bytes_of_each_launch_of_one_10M_row_pass": [x * 2048 for x in fv10] if fv10 else None, "pow2_rows_launch": pow2,
          "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, round-3 binaries) per %d-row launch (average of the passes' launches); FETCH_SIZE*1024*2 "
                  "(gfx950 half-count correction of wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE*1024; L2->fabric side: "
                  "requests the Infinity Cache serves are counted too (profiles/r02_mall has the probe); profiles/r03_headline" % rows}
    json.dump(tr, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    extra = {}
    for pref in ("rr_syrk_f32_diag16_kernel", "rr_rff_features_mfma_kernel"):
        f2, w2 = _per_dispatch("headline_fetch", pref, "FETCH_SIZE"), _per_dispatch("headline_write", pref, "WRITE_SIZE")
        if f2 and w2:
            extra[pref] = {"fetch_bytes_per_launch": max(f2) * 2048, "write_bytes_per_launch": max(w2) * 1024}
    p2 = os.path.join(ROOT, "profiles", "r03_headline", "summary.json")
    if os.path.exists(p2):
        sm = json.load(open(p2))
        sm["hbm_side_traffic"] = {"rr_syrk_f32_kernel": tr, **extra}
        json.dump(sm, open(p2, "w"), indent=1)
    for tag, name in (("headline_fetch", "pmc_fetch_pow2_rows.csv"), ("headline_write", "pmc_write.csv"), ("headline10m_fetch", "pmc_fetch.csv")):
      if os.path.exists(os.path.join(SRC, tag, "p_counter_collection.csv")):
        shutil.copy(os.path.join(SRC, tag, "p_counter_collection.csv"), os.path.join(ROOT, "profiles", "r03_headline", name))
    print("traffic.json", json.dumps(tr)[:300])
