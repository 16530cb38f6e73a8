#!/bin/bash
# A/B of rr_chol_diag_pipe_kernel (default) against rr_chol_diag_kernel (RR_CHOL_DIAG=0): posterior parity tests on both,
# then the posterior configurations of bench.py alternating on the same box.
out=gpurun_out/${1:-chol}
mkdir -p $out
T="tests/test_gpu_slm.py tests/test_gpu_rff.py::test_gram_posterior_weights tests/test_gpu_parity_r2.py tests/test_gpu_deterministic.py"
timeout 900 python -m pytest $T -q -m gpu -x > $out/pytest_pipe.log 2>&1; echo "pipe rc=$?"; tail -3 $out/pytest_pipe.log | cut -c1-300
RR_CHOL_DIAG=0 timeout 600 python -m pytest tests/test_gpu_slm.py -q -m gpu -x -k "posterior or posdef" > $out/pytest_plain.log 2>&1; echo "plain rc=$?"; tail -3 $out/pytest_plain.log | cut -c1-300
Q="--no-cpu-baseline --no-alt-engine --no-parity-check --rows 1000000 --steps 1 --warmup 0 --configs posterior_f4096,posterior_f8257"
for rep in 1 2; do
  RR_CHOL_DIAG=0 python bench.py $Q > $out/plain_$rep.json 2> $out/plain_$rep.err
  python bench.py $Q > $out/pipe_$rep.json 2> $out/pipe_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/p*_?.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1])["configs"]
        print(f, {k:(round(v["ms"],3), v.get("parity_vs_oracle_solve_posdef")) for k,v in d.items() if k.startswith("posterior")})
PY
