#!/bin/bash
# Round-3 GPU check: the new tests, then every new bench configuration on its own under a watchdog.
out=gpurun_out/${1:-r3b}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_parity_r2.py tests/test_gpu_rff.py -x -q > $out/pytest_new.log 2>&1
echo "pytest rc=$?"; tail -15 $out/pytest_new.log
for cfg in c2_elbo_eval c2f64_elbo_eval_n200k posterior_f4096 posterior_f8257 predict_moments_n300k c2laplace_f64phase_n1m; do
  timeout 420 python bench.py --rows 1000000 --steps 1 --warmup 1 --no-cpu-baseline --no-alt-engine --config-timeout 300 --configs $cfg > $out/$cfg.json 2> $out/$cfg.err
  echo "$cfg rc=$? $(tail -c 300 $out/$cfg.err | tr '\n' ' ')"
done
