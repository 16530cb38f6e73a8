#!/usr/bin/env python3
"""Kernel timeline of one steady-state GLM SVI step from a rocprofv3 --kernel-trace directory: start offset, idle gap
before each kernel, duration (step = from one rr_glm_grad_t_kernel to the next).  python tools/step_timeline.py <dir>"""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find index of a 'rr_glm_grad_t_kernel' late in the run, print until the next one
idx=[i for i,r in enumerate(rows) if 'rr_glm_grad_t_kernel' in r['Kernel_Name']]
a=idx[len(idx)//2]; b=idx[len(idx)//2+1]
t0=int(rows[a]['End_Timestamp']); prev=t0
for r in rows[a+1:b+1]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print("%8.1f us gap %7.1f dur %8.1f  %s" % ((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,r['Kernel_Name'][:50]))
    prev=e
print("step total %.1f us" % ((prev-t0)/1e3))
