#!/bin/bash
# buffer-descriptor, staggered LDS-DMA in every f32 tile kernel: parity, then the affected bench entries
out=gpurun_out/${1:-bufdma}
mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_rff.py tests/test_gpu_slm.py tests/test_gpu_glm.py tests/test_gpu_parity_r2.py tests/test_gpu_large_xdim.py -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?"; tail -4 $out/pytest.log | cut -c1-300
python bench.py --no-cpu-baseline --no-alt-engine --steps 5 --warmup 2 --configs c2_elbo_eval,c3,c5,predict_moments_n300k > $out/bench.json 2> $out/bench.err
python - <<PY
import json
l=[x for x in open("$out/bench.json") if x.startswith("{")]
d=json.loads(l[-1]); r=d["roofline"]
print("headline", round(d["value"]/1e6,3), round(d["ms_per_step"],1), "syrk", round(r["avg_launch_ms"],2), round(r["frac"],4), "both", round(r["gram_both_kernels_frac"],4), "whole", round(r["whole_path_frac"],4), d["config"].get("parity_rel_err_2048_rows_vs_oracle"))
for k,v in d["configs"].items():
    print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if isinstance(vv,(int,float))}, {kk:round(vv,4) for kk,vv in v.get("roofline",{}).items() if isinstance(vv,float)})
    if "elbo_eval_one_gpu_share" in v: print("   ", v["elbo_eval_one_gpu_share"])
    if "samplers" in v: print("   ", {kk:(round(vv["device_calls_ms"],3), round(vv["gemm_frac_over_device_calls"],4), round(vv["fit_step_ms"],3)) for kk,vv in v["samplers"].items()})
PY
