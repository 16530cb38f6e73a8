#!/bin/bash
out=gpurun_out/${1:-r3e}
mkdir -p $out
timeout 900 python -m pytest tests/test_debug_builds.py -q -m gpu > $out/pytest_debug.log 2>&1; echo "debug rc=$?"; tail -3 $out/pytest_debug.log
timeout 1200 python -m pytest tests/test_gpu_rff.py tests/test_gpu_slm.py tests/test_gpu_deterministic.py tests/test_gpu_parity_r2.py tests/test_gpu_glm.py -q -x > $out/pytest_sel.log 2>&1; echo "sel rc=$?"; tail -3 $out/pytest_sel.log
Q="--no-cpu-baseline --no-alt-engine --steps 3 --warmup 1 --configs none"
for rep in 1 2; do
  RR_SYRK_NO_DIAG16=1 python bench.py $Q > $out/diag_old_$rep.json 2> $out/diag_old_$rep.err
  python bench.py $Q > $out/diag16_$rep.json 2> $out/diag16_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/diag*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); r=d["roofline"]
        print(f, round(d["ms_per_step"],2), round(r["whole_path_frac"],4), round(r["frac"],4), r["other_kernels_ms_per_step"])
PY
timeout 600 python bench.py --rows 1000000 --steps 1 --warmup 1 --no-cpu-baseline --no-alt-engine --configs c5 > $out/c5.json 2> $out/c5.err; echo "c5 rc=$?"
