#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a tools/prof.sh run (gpurun_out/prof) into profiles/<tag>/ and
derive the per-launch HBM-side traffic of the dominant kernel into profiles/traffic.json.

    python tools/summarize_prof.py r01_v4

Traffic = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024 bytes: the counters are in KiB, come from
separate --pmc passes, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section).  The counters sit on the L2 -> fabric side and include
Infinity-Cache hits."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "rr_syrk_f32_kernel"  # dominant kernel (prefix match)
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(SRC, "kt", "kt_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "kt_bench.json"), os.path.join(dst, "bench_under_rocprof.json"))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in ("fetch", "write", "sq", "lds"):
    f = os.path.join(SRC, "pmc_%s" % p, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    shutil.copy(f, os.path.join(dst, "pmc_%s.csv" % p))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
summary = {k: dict(v) for k, v in agg.items() if k.startswith("rr_")}
k = summary.get(KERNEL, {})
rows = json.loads([l for l in open(os.path.join(SRC, "pmc_fetch.json")) if l.startswith("{")][-1])["roofline"]["rows_per_step"]
traffic = None
if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
    traffic = {"kernel": KERNEL, "rows_per_launch": rows,
               "fetch_bytes": k["FETCH_SIZE"] * 1024 * 2, "write_bytes": k["WRITE_SIZE"] * 1024,
               "hbm_bytes": k["FETCH_SIZE"] * 1024 * 2 + k["WRITE_SIZE"] * 1024,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of one %d-row launch; "
                       "FETCH_SIZE*1024*2 (gfx950 half-count correction) + WRITE_SIZE*1024; L2->fabric side, "
                       "includes Infinity-Cache hits; profiles/%s" % (rows, tag)}
    if KERNEL == "rr_syrk_f32_kernel":  # bench.py's default line quotes this file
        json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k:
    summary[KERNEL]["mfma_busy_frac"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (k["GRBM_GUI_ACTIVE"] / 8 * 1024)
json.dump({"counters": summary, "traffic": traffic}, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(open(os.path.join(dst, "kernel_stats.csv")).read())
print(json.dumps(traffic, indent=1))
print("mfma_busy_frac", summary.get(KERNEL, {}).get("mfma_busy_frac"))
print(json.dumps(summary.get(KERNEL, {}), indent=1))
