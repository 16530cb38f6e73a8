#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a tools/prof_r02.sh run (gpurun_out/prof_r02) into profiles/r02_<name>/ and derive,
per profile, a summary.json in which every hot kernel carries ROWS PER LAUNCH, its algorithmic bytes / flops per launch,
the achieved rate from rocprofv3's average duration and the fraction of the peak that bounds it.

    python tools/summarize_r02.py

HBM-side traffic = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024 bytes: counters are in KiB, come from separate --pmc passes,
and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); they sit
on the L2 -> fabric side and include Infinity-Cache hits."""
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r02")
PEAK = {"f32": 157.3e12, "f64": 78.6e12, "f16": 2500e12, "hbm": 8.0e12}


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


def kernel_trace(tag, prefix="kt"):
    """[(kernel name, duration ms)] in launch order."""
    p = os.path.join(SRC, tag, prefix + "_kernel_trace.csv")
    out = []
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            out.append((r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    return out


def counter_rows(tag):
    p = os.path.join(SRC, tag, "p_counter_collection.csv")
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []


def per_dispatch(tag, kernel_prefix, ctr):
    """Counter value of every dispatch of a kernel (summed over the counter's dimensions)."""
    vals = collections.defaultdict(float)
    for r in counter_rows(tag):
        if r["Kernel_Name"].replace("void ", "").startswith(kernel_prefix) and r["Counter_Name"] == ctr:
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return list(vals.values())


def find(stats, prefix):
    for k in stats:
        if k.replace("void ", "").startswith(prefix):
            return k
    return None


def put(name, files, summary):
    dst = os.path.join(ROOT, "profiles", name)
    os.makedirs(dst, exist_ok=True)
    for src, dname in files:
        if os.path.exists(src):
            shutil.copy(src, os.path.join(dst, dname))
    json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
    print(name, json.dumps(summary)[:500])


def hbm_bytes(tag_fetch, tag_write, prefix, full_only=False):
    fv, wv = per_dispatch(tag_fetch, prefix, "FETCH_SIZE"), per_dispatch(tag_write, prefix, "WRITE_SIZE")
    if not fv or not wv:
        return None
    if full_only:  # the full-size launches only
        fv = [x for x in fv if x > 0.5 * max(fv)]
        wv = [x for x in wv if x > 0.5 * max(wv)]
    return {"launches_counted": len(fv), "fetch_bytes_per_launch": sum(fv) / len(fv) * 1024 * 2,
            "write_bytes_per_launch": sum(wv) / len(wv) * 1024,
            "note": "FETCH_SIZE*1024*2 (gfx950 half-count of wide coalesced reads) and WRITE_SIZE*1024, separate --pmc passes; "
                    "L2->fabric side, Infinity-Cache hits included"}


def top_avg(durs, frac=0.97):
    """Average over the full-size launches: those within `frac` of the MEDIAN of the longer half (a first launch that
    also pays the code-object load is an outlier above it and is left out)."""
    durs = sorted(durs)
    ref = durs[len(durs) // 2:]
    ref = ref[len(ref) // 2]
    top = [d for d in durs if frac * ref < d < ref / frac]
    return sum(top) / len(top), len(top)


# ---- headline (N = 2M rows per launch) ----
b = bench_line("headline_kt")
if b:
    st = kernel_stats("headline_kt")
    rows = b["roofline"]["rows_per_step"] // max(b["roofline"]["launches_per_step"], 1)
    F = 4096
    off = b["roofline"]["flops_per_row"]
    dg = F * (F + 1.0) - off
    ks = {}
    for prefix, fl in (("rr_syrk_f32_kernel", off * rows), ("rr_syrk_f32_diag_kernel", dg * rows)):
        k = find(st, prefix + "(")
        ks[prefix] = dict(st[k], rows_per_launch=rows, algorithmic_flops_per_launch=fl,
                          achieved_tflops=fl / (st[k]["avg_ms"] * 1e-3) / 1e12,
                          frac_of_peak=fl / (st[k]["avg_ms"] * 1e-3) / PEAK["f32"], bound="f32 MFMA 157.3 TFLOP/s")
    k = find(st, "rr_rff_features_mfma_kernel")
    by = rows * (4.0 * 32 + 4.0 + 4.0 * F)
    ks["rr_rff_features_mfma_kernel"] = dict(st[k], rows_per_launch=rows, algorithmic_bytes_per_launch=by,
                                             achieved_GBs=by / (st[k]["avg_ms"] * 1e-3) / 1e9,
                                             frac_of_peak=by / (st[k]["avg_ms"] * 1e-3) / PEAK["hbm"], bound="HBM 8 TB/s (write)")
    tr = {p: hbm_bytes("headline_fetch", "headline_write", p) for p in ("rr_syrk_f32_kernel(", "rr_syrk_f32_diag_kernel(",
                                                                         "rr_rff_features_mfma_kernel")}
    sq = {}
    for ctr in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU_MFMA_MOPS_F32", "GRBM_GUI_ACTIVE"):
        v = per_dispatch("headline_sq", "rr_syrk_f32_kernel(", ctr)
        if v:
            sq[ctr] = sum(v) / len(v)
    if "GRBM_GUI_ACTIVE" in sq and "SQ_VALU_MFMA_BUSY_CYCLES" in sq:
        sq["mfma_busy_frac"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8 * 1024)
    put("r02_headline", [(os.path.join(SRC, "headline_kt", "kt_kernel_stats.csv"), "kernel_stats.csv"),
                         (os.path.join(SRC, "headline_kt.json"), "bench_under_rocprof.json"),
                         (os.path.join(SRC, "headline_fetch", "p_counter_collection.csv"), "pmc_fetch.csv"),
                         (os.path.join(SRC, "headline_write", "p_counter_collection.csv"), "pmc_write.csv"),
                         (os.path.join(SRC, "headline_sq", "p_counter_collection.csv"), "pmc_sq.csv")],
        {"command": "python bench.py --rows 2000000 --steps 3 --warmup 1 --configs none (tools/prof_r02.sh headline)",
         "kernels": ks, "hbm_side_traffic": tr, "syrk_f32_sq_counters_per_launch": sq})
    t = tr.get("rr_syrk_f32_kernel(")
    if t:
        json.dump({"kernel": "rr_syrk_f32_kernel", "rows_per_launch": rows, "fetch_bytes": t["fetch_bytes_per_launch"],
                   "write_bytes": t["write_bytes_per_launch"],
                   "hbm_bytes": t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"],
                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of one %d-row launch; FETCH_SIZE*1024*2 "
                           "(gfx950 half-count correction) + WRITE_SIZE*1024; L2->fabric side: requests the Infinity Cache "
                           "serves are counted too (no TCC counter separates them; profiles/r02_mall has the probe); "
                           "profiles/r02_headline" % rows},
                  open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)

# ---- configurations ----
NAMES = {"headline_shape_f64": "r02_f64", "C3_matern52_linear_concat_one_gpu_share": "r02_c3",
         "C4_fastfood_f16384": "r02_c4_fastfood", "C5_glm_poisson_svi_step": "r02_c5_glm"}
for tag, key in (("cfg_headline_kt", "headline_shape_f64"), ("cfg_c3_kt", "C3_matern52_linear_concat_one_gpu_share"),
                 ("cfg_c4_kt", "C4_fastfood_f16384"), ("cfg_c5_kt", "C5_glm_poisson_svi_step")):
    b = bench_line(tag)
    if not b or key not in b.get("configs", {}) or "error" in b["configs"][key]:
        continue
    cfg = b["configs"][key]
    st = kernel_stats(tag)
    trace = kernel_trace(tag)
    ks = {}
    if key == "headline_shape_f64":
        rows, F = cfg["rows_per_launch"], 4096
        # the small launches belong to the 4096-row parity check: quote the full-size launches from the trace
        # off-diagonal 128x128 tiles and the diagonal-tile kernel, each on its own algorithmic flops
        off_fl = (float(F) * F - 128.0 * F) * rows
        dg_fl = (128.0 * F + F) * rows
        tot = 0.0
        for prefix, fl in (("rr_syrk_f64_kernel(", off_fl), ("rr_syrk_f64_diag_kernel(", dg_fl)):
            durs = [d for n, d in trace if n.startswith(prefix)]
            if not durs:
                continue
            avg, nfull = top_avg(durs)
            tot += avg
            ks[prefix.rstrip("(")] = {"calls_full_size": nfull, "avg_ms": avg, "rows_per_launch": rows,
                                      "algorithmic_flops_per_launch": fl, "achieved_tflops": fl / (avg * 1e-3) / 1e12,
                                      "frac_of_peak": fl / (avg * 1e-3) / PEAK["f64"], "bound": "f64 MFMA 78.6 TFLOP/s",
                                      "all_calls": st[find(st, prefix)]}
        ks["both_f64_syrk_kernels"] = {"ms_per_launch": tot, "rows_per_launch": rows,
                                       "algorithmic_flops_per_launch": F * (F + 1.0) * rows,
                                       "achieved_tflops": F * (F + 1.0) * rows / (tot * 1e-3) / 1e12,
                                       "frac_of_peak": F * (F + 1.0) * rows / (tot * 1e-3) / PEAK["f64"]}
        favg, nf = top_avg([d for n, d in trace if "rr_rff_features_mfma64_kernel" in n or "rr_rff_features_kernel" in n])
        by = rows * (8.0 * 32 + 8.0 + 8.0 * F)
        ks["rr_rff_features_mfma64_kernel"] = {"calls_full_size": nf, "avg_ms": favg, "rows_per_launch": rows,
                                               "algorithmic_bytes_per_launch": by, "achieved_GBs": by / (favg * 1e-3) / 1e9,
                                               "frac_of_peak": by / (favg * 1e-3) / 1e9 / 8000.0,
                                               "bound": "HBM 8 TB/s (write); projection on the f64 MFMA, sin / cos on the f64 VALU"}
    elif key.startswith("C3"):
        rows, F = cfg["rows_per_launch"], 8257
        nb = (F + 255) // 256
        w = [min(256, F - 256 * i) for i in range(nb)]
        off = 2.0 * sum(w[i] * w[j] for i in range(nb) for j in range(i + 1, nb))
        dg = float(sum(x * (x + 1) for x in w))
        wl = F - 256 * (nb - 1)                      # valid columns of the ragged last block
        rag = 2.0 * 256 * (nb - 1) * wl if wl <= 192 else 0.0   # its off-diagonal tiles: rr_syrk_f32_ragged_kernel
        for prefix, fl in (("rr_syrk_f32_kernel(", off - rag), ("rr_syrk_f32_ragged_kernel(", rag), ("rr_syrk_f32_diag_kernel(", dg)):
            if not any(n.startswith(prefix) for n, _ in trace):
                continue
            # launches over the full 254 200-row chunks (the last chunk of a pass is the remainder)
            avg, nfull = top_avg([d for n, d in trace if n.startswith(prefix)])
            ks[prefix.rstrip("(")] = {"calls_full_chunk": nfull, "avg_ms": avg, "rows_per_launch": rows,
                                      "algorithmic_flops_per_launch": fl * rows,
                                      "achieved_tflops": fl * rows / (avg * 1e-3) / 1e12,
                                      "frac_of_peak": fl * rows / (avg * 1e-3) / PEAK["f32"], "bound": "f32 MFMA 157.3 TFLOP/s",
                                      "all_calls": st[find(st, prefix)]}
    elif key.startswith("C4"):
        rows = cfg["roofline"]["rows_per_launch"]
        avg, nfull = top_avg([d for n, d in trace if "rr_fastfood16_kernel" in n], 0.5)
        by = cfg["roofline"]["bytes_per_row"] * rows
        ks["rr_fastfood16_kernel"] = {"calls_full_chunk": nfull, "avg_ms": avg, "rows_per_launch": rows,
                                      "algorithmic_bytes_per_launch": by, "achieved_GBs": by / (avg * 1e-3) / 1e9,
                                      "frac_of_peak": by / (avg * 1e-3) / PEAK["hbm"], "bound": "HBM 8 TB/s (write)",
                                      "all_calls": st[find(st, "rr_fastfood16_kernel")]}
        t = hbm_bytes("cfg_c4_fetch", "cfg_c4_write", "rr_fastfood16_kernel", full_only=True)
        if t:
            t.update(algorithmic_read_bytes=4.0 * 128 * rows, algorithmic_write_bytes=4.0 * 16384 * rows)
            ks["rr_fastfood16_kernel"]["hbm_side_traffic"] = t
    else:
        M, F, K, L = 65536, 2048, 10, 50
        k = find(st, "rr_gemm_tn_f32_kernel")
        fl = 2.0 * K * L * M * F
        ks["rr_gemm_tn_f32_kernel"] = dict(st[k], rows_per_launch=M, algorithmic_flops_per_launch=fl,
                                           achieved_tflops=fl / (st[k]["avg_ms"] * 1e-3) / 1e12,
                                           frac_of_peak=fl / (st[k]["avg_ms"] * 1e-3) / PEAK["f32"],
                                           bound="f32 MFMA 157.3 TFLOP/s; (K L = 500) x 65536 x 2048 per GEMM, 3 per step")
        for p in ("rr_glm_lik_kernel", "rr_glm_grad_t_kernel", "rr_transpose_f32_kernel", "rr_glm_reduce_kernel",
                  "rr_glm_draw_kernel", "rr_gather_rows_kernel", "rr_rff_features_mfma_kernel<32, 4, false"):
            kk = find(st, p)
            if kk:
                ks[p] = dict(st[kk], rows_per_launch=M)
        steps = st[k]["calls"] / 3.0
        ks["steps_profiled"] = steps
        ks["kernel_ms_per_step"] = sum(v["total_ms"] for kk, v in st.items()
                                       if any(s in kk for s in ("glm", "gemm_tn", "transpose", "gather")) or
                                       "features_mfma_kernel<32, 4, false" in kk) / steps
    files = [(os.path.join(SRC, tag, "kt_kernel_stats.csv"), "kernel_stats.csv"),
             (os.path.join(SRC, tag + ".json"), "bench_under_rocprof.json")]
    if key.startswith("C4"):
        files += [(os.path.join(SRC, "cfg_c4_fetch", "p_counter_collection.csv"), "pmc_fetch.csv"),
                  (os.path.join(SRC, "cfg_c4_write", "p_counter_collection.csv"), "pmc_write.csv")]
    put(NAMES[key], files,
        {"command": "python bench.py --rows 500000 --steps 1 --warmup 0 --configs %s (tools/prof_r02.sh configs)" % tag.split("_")[1],
         "config": {k2: v for k2, v in cfg.items() if k2 != "cpu_baseline"}, "kernels": ks})

# ---- Infinity-Cache probe: is the SYRK's L2->fabric fetch served by HBM or by the MALL? ----
mall = {}
for rows_l in (16384, 2000000):
    tag = "mall_%d_fetch" % rows_l
    fv = per_dispatch(tag, "rr_syrk_f32_kernel(", "FETCH_SIZE")
    tr_ = [d for n, d in kernel_trace(tag, "p") if n.startswith("rr_syrk_f32_kernel(")]
    if fv:
        nl = len(fv)
        mall[str(rows_l)] = {"rows_per_launch": rows_l, "launches": nl, "P_bytes_per_launch": rows_l * 4096 * 4,
                             "fetch_bytes_per_row": sum(fv) * 1024 * 2 / (nl * rows_l),
                             "kernel_us_per_1000_rows_under_pmc": sum(tr_) / len(tr_) * 1e3 / (rows_l / 1000.0) if tr_ else None}
if mall:
    dst = os.path.join(ROOT, "profiles", "r02_mall")
    os.makedirs(dst, exist_ok=True)
    for rows_l in (16384, 2000000):
        src = os.path.join(SRC, "mall_%d_fetch" % rows_l, "p_counter_collection.csv")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(dst, "pmc_fetch_%d_rows_per_launch.csv" % rows_l))
    json.dump({"command": "RR_GRAM_CHUNK_ROWS=<rows> python bench.py --rows 2000000 --steps 1 --warmup 0 --configs none under "
                          "rocprofv3 --pmc FETCH_SIZE (tools/prof_r02.sh mall)",
               "question": "rr_syrk_f32_kernel requests ~5x the feature matrix from the fabric side of L2 per launch. Is that HBM "
                           "traffic or Infinity-Cache traffic?  A 16 384-row launch keeps its whole P (256 MiB) inside the 256 MiB "
                           "MALL; a 2M-row launch streams 32.8 GB through it.",
               "launch_sizes": mall}, open(os.path.join(dst, "summary.json"), "w"), indent=1)
    print("r02_mall", json.dumps(mall))

# ---- the opt-in fp16x3 engine ----
b = bench_line("fp16x3_kt")
if b:
    st = kernel_stats("fp16x3_kt")
    rows = b["roofline"]["rows_per_step"] // max(b["roofline"]["launches_per_step"], 1)
    k = find(st, "rr_syrk_b16w4_kernel")
    F = 4096
    ntile = 16 * 17 // 2
    issued = 3 * 2.0 * 65536 * ntile * rows
    t = hbm_bytes("fp16x3_fetch", "fp16x3_write", "rr_syrk_b16w4_kernel")
    sq = {}
    for ctr in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                "GRBM_GUI_ACTIVE"):
        v = per_dispatch("fp16x3_sq", "rr_syrk_b16w4_kernel", ctr)
        if v:
            sq[ctr] = sum(v) / len(v)
    if "GRBM_GUI_ACTIVE" in sq and k:
        sq["mfma_busy_frac"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8 * 1024)
        pd = [d for n, d in kernel_trace("fp16x3_sq", "p") if "rr_syrk_b16w4_kernel" in n]
        sq["avg_clock_GHz_under_pmc"] = sq["GRBM_GUI_ACTIVE"] / 8 / (sum(pd) / len(pd) * 1e-3) / 1e9 if pd else None
    put("r02_fp16x3", [(os.path.join(SRC, "fp16x3_kt", "kt_kernel_stats.csv"), "kernel_stats.csv"),
                       (os.path.join(SRC, "fp16x3_kt.json"), "bench_under_rocprof.json"),
                       (os.path.join(SRC, "fp16x3_fetch", "p_counter_collection.csv"), "pmc_fetch.csv"),
                       (os.path.join(SRC, "fp16x3_write", "p_counter_collection.csv"), "pmc_write.csv"),
                       (os.path.join(SRC, "fp16x3_sq", "p_counter_collection.csv"), "pmc_sq.csv")],
        {"command": "python bench.py --rows 2000000 --steps 3 --warmup 1 --configs none --engine fp16x3 (tools/prof_r02.sh engine)",
         "kernels": {"rr_syrk_b16w4_kernel": dict(st[k], rows_per_launch=rows, issued_flops_per_launch=issued,
                                                  issued_tflops=issued / (st[k]["avg_ms"] * 1e-3) / 1e12,
                                                  issued_frac_of_fp16_peak=issued / (st[k]["avg_ms"] * 1e-3) / PEAK["f16"],
                                                  algorithmic_tflops=F * (F + 1.0) * rows / (st[k]["avg_ms"] * 1e-3) / 1e12,
                                                  hbm_side_traffic=t, P_bytes_per_launch=rows * F * 4.0,
                                                  traffic_over_P=(t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]) / (rows * F * 4.0) if t else None)},
         "sq_counters_per_launch": sq})


# ---- the default command (N = 10M) ----
b = bench_line("default_kt")
if b:
    st = kernel_stats("default_kt")
    rows = b["roofline"]["rows_per_step"] // max(b["roofline"]["launches_per_step"], 1)
    k = find(st, "rr_syrk_f32_kernel(")
    fl = b["roofline"]["flops_per_row"] * rows
    put("r02_default10m", [(os.path.join(SRC, "default_kt", "kt_kernel_stats.csv"), "kernel_stats.csv"),
                           (os.path.join(SRC, "default_kt.json"), "bench_under_rocprof.json")],
        {"command": "python bench.py --no-cpu-baseline --no-alt-engine --no-parity-check --configs none (tools/prof_r02.sh default)",
         "kernels": {"rr_syrk_f32_kernel": dict(st[k], rows_per_launch=rows, algorithmic_flops_per_launch=fl,
                                                achieved_tflops=fl / (st[k]["avg_ms"] * 1e-3) / 1e12,
                                                frac_of_peak=fl / (st[k]["avg_ms"] * 1e-3) / PEAK["f32"])},
         "bench_line_avg_launch_ms": b["roofline"]["avg_launch_ms"], "bench_line_frac": b["roofline"]["frac"]})
