#!/bin/bash
# round 3: the GLM step's EdPhi product fused with its contraction -- parity tests, then config 5 with and without it
out=gpurun_out/${1:-glmfuse}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_glm.py -q -m gpu -x > $out/pytest_glm.log 2>&1; echo "glm rc=$?"; tail -5 $out/pytest_glm.log | cut -c1-400
Q="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs c5_glm_poisson_svi_step"
for rep in 1 2; do
  RR_GLM_NO_FUSE=1 timeout 300 python bench.py $Q > $out/plain_$rep.json 2> $out/plain_$rep.err
  RR_GLM_FUSE_LIK=0 timeout 300 python bench.py $Q > $out/fuseg_$rep.json 2> $out/fuseg_$rep.err
  timeout 300 python bench.py $Q > $out/fused_$rep.json 2> $out/fused_$rep.err
  RR_GEMM_SPLIT_ROUNDS=1 timeout 300 python bench.py $Q > $out/fusedr1_$rep.json 2> $out/fusedr1_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*_[12].json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1])["configs"]["C5_glm_poisson_svi_step"]["samplers"]
        print(f, {k:(round(v["device_calls_ms"],3), round(v["gemm_frac_over_device_calls"],4), round(v["fit_step_ms"],3), round(v["elbo_step_ms"],3)) for k,v in d.items()})
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o c5 -- python $GRAFT_REPO_ROOT/bench.py $Q > $GRAFT_REPO_ROOT/$out/prof_bench.json 2> $GRAFT_REPO_ROOT/$out/prof.err
cd $GRAFT_REPO_ROOT
ls $out/prof | head; f=$(ls $out/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 $f | cut -c1-200
