#!/bin/bash
# k-loop ablation of rr_syrk_f32_kernel (timing only: results are garbage): what do the in-loop LDS-DMA and the barrier cost?
out=gpurun_out/${1:-ablate}
mkdir -p $out
Q="--no-cpu-baseline --no-alt-engine --no-parity-check --rows 2000000 --steps 3 --warmup 1 --configs none"
for a in 0 1 2 3 0 1 2 3; do
  RR_GRAM_ABLATE=$a python bench.py $Q > $out/a${a}_$RANDOM.json 2> $out/err.log
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/a*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if not l: print(f,"NO"); continue
    d=json.loads(l[-1]); r=d["roofline"]
    print(f, "syrk ms/launch", round(r["avg_launch_ms"],2), "frac", round(r["frac"],4))
PY
