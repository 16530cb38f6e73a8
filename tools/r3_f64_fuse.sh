#!/bin/bash
out=gpurun_out/${1:-f64fuse}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_slm.py tests/test_gpu_parity_r2.py -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?"; tail -6 $out/pytest.log | cut -c1-400
Q="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs c2f64_elbo_eval_n200k"
for rep in 1 2; do
  RR_PASS2_NO_FUSE=1 python bench.py $Q > $out/plain_$rep.json 2> $out/plain_$rep.err
  python bench.py $Q > $out/fused_$rep.json 2> $out/fused_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*_[12].json")):
    l=[x for x in open(f) if x.startswith("{")]
    v=json.loads(l[-1])["configs"]["C2f64_elbo_eval_n200k"]
    print(f, v.get("ms"), {k:round(x,4) for k,x in v["roofline"].items() if isinstance(x,float)}, {k:x for k,x in v.items() if k.startswith("parity")})
PY
