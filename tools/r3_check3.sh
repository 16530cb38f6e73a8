#!/bin/bash
out=gpurun_out/${1:-r3h}
mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_rff.py tests/test_gpu_slm.py tests/test_gpu_deterministic.py tests/test_gpu_gram_engines.py tests/test_debug_builds.py -q -m gpu > $out/pytest_sel.log 2>&1; echo "sel rc=$?"; tail -4 $out/pytest_sel.log | cut -c1-300
Q="--no-cpu-baseline --no-alt-engine --steps 3 --warmup 1 --configs none"
for rep in 1 2; do
  RR_SYRK_DIAG_KB=32 python bench.py $Q > $out/kb32_$rep.json 2> $out/kb32_$rep.err
  python bench.py $Q > $out/kb64_$rep.json 2> $out/kb64_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/kb*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); r=d["roofline"]
        print(f, round(d["ms_per_step"],2), round(r["whole_path_frac"],4), round(r["frac"],4), r["other_kernels_ms_per_step"])
PY
