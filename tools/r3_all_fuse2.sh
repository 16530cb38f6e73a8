#!/bin/bash
out=gpurun_out/${1:-allfuse2}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_slm.py tests/test_gpu_parity_r2.py tests/test_gpu_glm.py -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?"; tail -6 $out/pytest.log | cut -c1-400
RR_PASS2_NO_FUSE=1 python tools/diag_c1b.py 2>&1 | tail -4; python tools/diag_c1b.py 2>&1 | tail -4
Q5="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs c5_glm_poisson_svi_step"
Q3="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs c3"
timeout 300 python bench.py $Q5 > $out/c5fused_1.json 2> $out/c5fused_1.err
RR_PASS2_NO_FUSE=1 timeout 600 python bench.py $Q3 > $out/c3plain_1.json 2> $out/c3plain_1.err
timeout 600 python bench.py $Q3 > $out/c3fused_1.json 2> $out/c3fused_1.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/c*_[12].json")):
    l=[x for x in open(f) if x.startswith("{")]
    if not l: print(f, "NO LINE"); continue
    d=json.loads(l[-1])["configs"]
    for k,v in d.items():
        if "samplers" in v:
            print(f, {kk:(round(vv["device_calls_ms"],3), round(vv["gemm_frac_over_device_calls"],4), round(vv["fit_step_ms"],3)) for kk,vv in v["samplers"].items()})
        else:
            print(f, k, json.dumps(v.get("elbo_eval_one_gpu_share")))
PY
