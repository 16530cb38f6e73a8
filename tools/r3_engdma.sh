#!/bin/bash
out=gpurun_out/${1:-engdma}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_gram_engines.py -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?"; tail -2 $out/pytest.log | cut -c1-300
for eng in fp16x3 bf16x3; do
python bench.py --no-cpu-baseline --engine $eng --steps 3 --warmup 1 --configs none > $out/$eng.json 2> $out/$eng.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if not l: print(f,"NO"); continue
    d=json.loads(l[-1]); r=d["roofline"]
    print(f, round(d["value"]/1e6,3), round(d["ms_per_step"],1), {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if isinstance(v,(int,float))})
PY
