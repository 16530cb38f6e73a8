#!/bin/bash
# A/B of the triangular k-range of C = Y^T Y (Syrk64Args::lower_tri, default) against RR_SYRK64_TRI=0: posterior parity tests, then the posterior
# configurations of bench.py alternating on the same box.
out=gpurun_out/${1:-tri}
mkdir -p $out
T="tests/test_gpu_slm.py tests/test_gpu_rff.py::test_gram_posterior_weights tests/test_gpu_parity_r2.py tests/test_gpu_deterministic.py tests/test_debug_builds.py"
timeout 900 python -m pytest $T -q -m gpu -x > $out/pytest_tri.log 2>&1; echo "tri rc=$?"; tail -3 $out/pytest_tri.log | cut -c1-300
Q="--no-cpu-baseline --no-alt-engine --no-parity-check --rows 1000000 --steps 1 --warmup 0 --configs posterior_f4096,posterior_f8257"
for rep in 1 2; do
  RR_SYRK64_TRI=0 python bench.py $Q > $out/plain_$rep.json 2> $out/plain_$rep.err
  python bench.py $Q > $out/tri_$rep.json 2> $out/tri_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*_?.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1])["configs"]
        print(f, {k:round(v["ms"],3) for k,v in d.items() if k.startswith("posterior")})
PY
