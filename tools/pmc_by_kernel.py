#!/usr/bin/env python3
"""Per kernel of a `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES ...` run
(tools/prof.sh sq / predictsq / c5sq): launches, average duration, matrix-pipe busy fraction on the CUs that had work,
clock, and the fraction of CU-cycles with work (SQ_BUSY_CU_CYCLES over 256 CUs x the launch's cycles).
    python tools/pmc_by_kernel.py <p_counter_collection.csv> [top n]"""
import collections, csv, sys
rows = csv.DictReader(open(sys.argv[1]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt, dur, seen = collections.Counter(), collections.defaultdict(float), set()
for r in rows:
    k = r["Kernel_Name"][:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        cnt[k] += 1
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:top]:
    cu = v["SQ_BUSY_CU_CYCLES"]
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * cu) if cu else 0.0
    clk = v["GRBM_GUI_ACTIVE"] / 8 / dur[k] if dur[k] else 0.0
    cuf = cu / (v["GRBM_GUI_ACTIVE"] / 8 * 256) if v["GRBM_GUI_ACTIVE"] else 0.0
    print("%-64s n=%4d avg %8.3f ms  mfma_busy %.3f  clock %.3f GHz  CUs_with_work %.3f" % (k, cnt[k], dur[k] / cnt[k] / 1e6, busy, clk, cuf))
