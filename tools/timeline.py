#!/usr/bin/env python3
"""Timeline (kernels + memory copies, ms relative to the window's start) around the LAST launch of a kernel, from a
`rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` directory:
    python tools/timeline.py <dir> <kernel name substring> [ms before] [ms after]"""
import csv, glob, sys
d, key = sys.argv[1], sys.argv[2]
before, after = (float(sys.argv[3]) if len(sys.argv) > 3 else 10.0), (float(sys.argv[4]) if len(sys.argv) > 4 else 3.0)
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s" % (r.get("Direction", ""), r.get("Bytes", r.get("Size", "")))))
ev.sort()
last = max(i for i, e in enumerate(ev) if key in e[2])
t0 = ev[last][0] - before * 1e6
for s, e, n in ev:
    if t0 <= s <= ev[last][1] + after * 1e6:
        print("%9.3f %9.3f  %8.3f ms  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
