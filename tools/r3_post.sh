#!/bin/bash
# posterior parity tests, then the posterior configurations of bench.py (twice)
out=gpurun_out/${1:-post}
mkdir -p $out
T="tests/test_gpu_slm.py tests/test_gpu_rff.py::test_gram_posterior_weights tests/test_gpu_parity_r2.py tests/test_gpu_deterministic.py tests/test_debug_builds.py"
timeout 900 python -m pytest $T -q -m gpu -x > $out/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $out/pytest.log | cut -c1-300
Q="--no-cpu-baseline --no-alt-engine --rows 1000000 --steps 1 --warmup 0 --configs posterior_f4096,posterior_f8257,c2f64_elbo_eval_n200k"
for rep in 1 2; do python bench.py $Q > $out/post_$rep.json 2> $out/post_$rep.err; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/post_?.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1])["configs"]
        print(f, {k:(round(v["ms"],3), v.get("parity_vs_oracle_solve_posdef")) for k,v in d.items() if k.startswith("posterior")})
        for k,v in d.items():
            if "elbo" in k: print(k, v.get("ms"), {a:b for a,b in v.items() if a.startswith("parity")})
PY
