#!/bin/bash
out=gpurun_out/${1:-c5order}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_glm.py tests/test_gpu_gram_engines.py -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?"; tail -3 $out/pytest.log | cut -c1-300
B="--no-alt-engine --rows 1000000 --steps 1 --warmup 0"
python bench.py $B --configs c5 > $out/a_c5_cpu.json 2> $out/a.err
python bench.py $B --no-cpu-baseline --configs c4,c5 > $out/b_c4c5.json 2> $out/b.err
RR_BENCH_C5_ORDER=device,host python bench.py $B --no-cpu-baseline --configs c4,c5 > $out/c_c4c5_swapped.json 2> $out/c.err
python bench.py $B --configs c3,c4,c5 > $out/d_c3c4c5_cpu.json 2> $out/d.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if not l: print(f,"NO"); continue
    v=json.loads(l[-1])["configs"]["C5_glm_poisson_svi_step"]["samplers"]
    print(f, {kk:(round(vv["device_calls_ms"],3), round(vv["gemm_frac_over_device_calls"],4), round(vv["fit_step_ms"],3), round(vv["elbo_step_ms"],2)) for kk,vv in v.items()})
PY
