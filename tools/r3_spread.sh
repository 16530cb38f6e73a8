#!/bin/bash
out=gpurun_out/${1:-spread}
mkdir -p $out
Q="--no-cpu-baseline --no-alt-engine --rows 2000000 --steps 3 --warmup 1 --configs c3,c2_elbo_eval,c5"
for sp in 2 1 0; do
  RR_DMA_SPREAD=$sp python bench.py $Q > $out/sp$sp.json 2> $out/sp$sp.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/sp*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if not l: print(f,"NO"); continue
    d=json.loads(l[-1]); r=d["roofline"]
    c=d["configs"]
    e3=c["C3_matern52_linear_concat_one_gpu_share"]["elbo_eval_one_gpu_share"]["ms"]
    print(f, "syrk", round(r["avg_launch_ms"],2), round(r["frac"],4), "| C3 gram", round(e3["statistics_pass"],1), "p2", round(e3["second_pass"],1), "| C2 p2", round(c["C2_elbo_eval"]["ms"]["second_pass"],1), "| C5", {kk:round(vv["device_calls_ms"],3) for kk,vv in c["C5_glm_poisson_svi_step"]["samplers"].items()})
PY
