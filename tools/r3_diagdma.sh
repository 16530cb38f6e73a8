#!/bin/bash
out=gpurun_out/${1:-diagdma}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_rff.py tests/test_gpu_slm.py tests/test_gpu_parity_r2.py -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?"; tail -2 $out/pytest.log | cut -c1-300
Q="--no-cpu-baseline --no-alt-engine --steps 3 --warmup 1 --configs none"
for rep in 1 2; do
 for sp in 2 0; do
  RR_DMA_SPREAD=$sp python bench.py $Q > $out/sp${sp}_$rep.json 2> $out/err.log
 done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/sp*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if not l: print(f,"NO"); continue
    d=json.loads(l[-1]); r=d["roofline"]
    print(f, round(d["value"]/1e6,3), round(d["ms_per_step"],1), "syrk", round(r["avg_launch_ms"],2), round(r["frac"],4), "both", round(r["gram_both_kernels_frac"],4), "whole", round(r["whole_path_frac"],4), r["other_kernels_ms_per_step"])
PY
