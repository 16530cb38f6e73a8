#!/bin/bash
# rr_syrk_f32_kernel's LDS-DMA: flat / buffer-descriptor requests, right after the barrier / staggered over the first k-step
# pairs (RR_SYRK_STAGGER = 0 default, 1 flat staggered, 2 buffer, 3 buffer staggered): parity, then same-box A/B
out=gpurun_out/${1:-stag}
mkdir -p $out
for m in 2 3; do
RR_SYRK_STAGGER=$m timeout 900 python -m pytest tests/test_gpu_rff.py tests/test_gpu_slm.py -q -m gpu -k "gram or elbo or fit" > $out/pytest_$m.log 2>&1; echo "mode $m rc=$?"; tail -1 $out/pytest_$m.log | cut -c1-300
done
Q="--no-cpu-baseline --no-alt-engine --rows 2000000 --steps 3 --warmup 1 --configs none"
for rep in 1 2; do
  for m in 0 2 3; do
  RR_SYRK_STAGGER=$m python bench.py $Q > $out/m${m}_$rep.json 2> $out/m${m}_$rep.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/m*_[12].json")):
    l=[x for x in open(f) if x.startswith("{")]
    if not l: print(f,"NO"); continue
    d=json.loads(l[-1]); r=d["roofline"]
    print(f, "syrk", round(r["avg_launch_ms"],2), round(r["frac"],4), "whole", round(r["whole_path_frac"],4), "parity", d["config"].get("parity_rel_err_2048_rows_vs_oracle"), d["config"].get("trace_rel_err"))
PY
