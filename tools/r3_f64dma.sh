#!/bin/bash
out=gpurun_out/${1:-f64dma}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_rff.py tests/test_gpu_slm.py tests/test_gpu_parity_r2.py tests/test_gpu_deterministic.py -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?"; tail -2 $out/pytest.log | cut -c1-300
Q="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs headline_shape_f64,c2f64_elbo_eval_n200k,posterior_f4096,posterior_f8257"
for rep in 1 2; do python bench.py $Q > $out/r$rep.json 2> $out/err.log; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/r*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    d=json.loads(l[-1])["configs"]
    for k,v in d.items(): print(f, k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if isinstance(vv,(int,float))}, {kk:round(vv,4) for kk,vv in v.get("roofline",{}).items() if isinstance(vv,float)}, v.get("ms"))
PY
