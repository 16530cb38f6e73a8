#!/bin/bash
# round 3: concatenated second pass with planned children -- parity tests, then config 3's evaluation A/B
out=gpurun_out/${1:-c3fuse}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_slm.py tests/test_gpu_parity_r2.py tests/test_gpu_glm.py -q -m gpu -x > $out/pytest.log 2>&1; echo "rc=$?"; tail -5 $out/pytest.log | cut -c1-400
Q="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs c3"
for rep in 1; do
  RR_PASS2_NO_FUSE=1 timeout 600 python bench.py $Q > $out/plain_$rep.json 2> $out/plain_$rep.err
  timeout 600 python bench.py $Q > $out/fused_$rep.json 2> $out/fused_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*_[12].json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1])["configs"]
        for k,v in d.items():
            print(f, k, {kk:vv for kk,vv in v.items() if isinstance(vv,(int,float))})
            print("   ", {kk:vv for kk,vv in v.get("roofline",{}).items() if isinstance(vv,(int,float))})
PY
