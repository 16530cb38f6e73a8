#!/bin/bash
# round 3: fused products with rotating epilogue register sets -- parity tests, then A/B of config 5, config 2's evaluation, config 3
out=gpurun_out/${1:-allfuse}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_slm.py tests/test_gpu_parity_r2.py tests/test_gpu_glm.py -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?"; tail -6 $out/pytest.log | cut -c1-400
python tools/diag_c1.py 2>&1 | tail -3
Q5="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs c5_glm_poisson_svi_step"
Q2="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs c2_elbo_eval"
Q3="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs c3"
for rep in 1 2; do
  RR_GLM_NO_FUSE=1 timeout 300 python bench.py $Q5 > $out/c5plain_$rep.json 2> $out/c5plain_$rep.err
  timeout 300 python bench.py $Q5 > $out/c5fused_$rep.json 2> $out/c5fused_$rep.err
done
RR_PASS2_NO_FUSE=1 timeout 300 python bench.py $Q2 > $out/c2plain_1.json 2> $out/c2plain_1.err
timeout 300 python bench.py $Q2 > $out/c2fused_1.json 2> $out/c2fused_1.err
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
            print(f, k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if isinstance(vv,(int,float))})
            print("   ", {kk:round(vv,4) for kk,vv in v.get("roofline",{}).items() if isinstance(vv,(int,float))})
PY
