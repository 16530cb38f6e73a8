#!/bin/bash
out=gpurun_out/${1:-predfuse}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_slm.py tests/test_gpu_parity_r2.py tests/test_gpu_rff.py tests/test_gpu_large_xdim.py -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?"; tail -4 $out/pytest.log | cut -c1-300
Q="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs predict_moments_n300k"
for rep in 1 2; do
  RR_PREDICT_NO_FUSE=1 python bench.py $Q > $out/plain_$rep.json 2> $out/plain_$rep.err
  python bench.py $Q > $out/fused_$rep.json 2> $out/fused_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*_[12].json")):
    l=[x for x in open(f) if x.startswith("{")]
    v=json.loads(l[-1])["configs"]["predict_moments_n300k"]
    print(f, round(v["ms"],3), round(v["roofline"]["frac"],4), v.get("parity_512_rows_vs_oracle"))
PY
