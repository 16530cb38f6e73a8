#!/bin/bash
# Same-box A/B runs of bench.py under environment switches (every gpurun call lands on a different box, and boxes differ by
# 1-2 %: only runs of ONE call compare).  One script for every measurement switch of DESIGN.md 8.1 / docs/KERNELS.md:
#
#   bash tools/ab.sh <name> "<arm>|<arm>|..." "<bench.py arguments>" [reps] [pytest selection run under every arm first]
#
# an arm is a list of VAR=value settings ("" = the default build); arms alternate rep by rep; every run's unabridged record
# goes to gpurun_out/ab_<name>/<arm index>_<rep>.json and a table of the numbers that matter is printed at the end.
#
# Recipes (name: arms; bench arguments) -- what rounds 3-4 measured with it:
#   stagger    "RR_SYRK_STAGGER=0|RR_SYRK_STAGGER=2|"                 "--rows 2000000 --steps 3 --warmup 1 --configs none"
#   spread     "RR_DMA_SPREAD=0|RR_DMA_SPREAD=1|"                     "--steps 3 --warmup 1 --configs none"
#   ablate     "RR_GRAM_ABLATE=1|RR_GRAM_ABLATE=2|RR_GRAM_ABLATE=3|"  "--no-parity-check --rows 2000000 --steps 3 --warmup 1 --configs none"
#   diag16     "RR_SYRK_NO_DIAG16=1|RR_SYRK_DIAG_KB=64|"              "--steps 3 --warmup 1 --configs none"
#   chunks     "RR_GRAM_CHUNK_ROWS=2097152|"                          "--steps 3 --warmup 1 --configs none"
#   overlap    "RR_GRAM_OVERLAP=1|"                                   "--steps 3 --warmup 1 --configs none"
#   fuse       "RR_PASS2_NO_FUSE=1|"                                  "--rows 1000000 --steps 1 --warmup 0 --configs c2_elbo_eval,c3"
#   glmfuse    "RR_GLM_NO_FUSE=1|RR_GLM_FUSE_LIK=0|"                  "--rows 1000000 --steps 1 --warmup 0 --configs c5"
#   c5order    "RR_BENCH_C5_ORDER=device,host|"                       "--rows 1000000 --steps 1 --warmup 0 --configs c4,c5"
#   predict    "RR_PREDICT_NO_FUSE=1|"                                "--rows 1000000 --steps 1 --warmup 0 --configs predict_moments_n300k"
#   posterior  "RR_POSDEF_LOOKAHEAD=0|RR_GEMM64_K128=0|RR_SYRK64_TRI=0|RR_CHOL_DIAG=0|RR_POSDEF_OVERLAP=0|" \
#              "--no-parity-check --rows 1000000 --steps 1 --warmup 0 --configs posterior_f4096,posterior_f8257,posterior_f16384"
#   fastfood   "RR_FASTFOOD_FIT=dense|"                               "--rows 1000000 --steps 1 --warmup 0 --configs c4elbo"
#   mergediag  "RR_SYRK_MERGE_DIAG=1|"                                "--steps 3 --warmup 1 --configs none"   (profiles/r04_diag)
#   predictpair "RR_PREDICT_NO_PAIR=1|"                               "--rows 1000000 --steps 1 --warmup 0 --configs predict_moments_n300k"
#   det        "RR_DETERMINISTIC=1|"                                  "--steps 3 --warmup 1 --configs c2_elbo_eval"
name=${1:?name}; arms=${2:?arms}; args=${3:?bench arguments}; reps=${4:-2}; tests=$5
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/ab_$name
mkdir -p $out
IFS='|' read -r -a ARMS <<< "$arms|"
if [ -n "$tests" ]; then
  for i in "${!ARMS[@]}"; do
    env ${ARMS[$i]} timeout 1500 python -m pytest $tests -q -m gpu -x > $out/pytest_$i.log 2>&1; echo "arm $i [${ARMS[$i]:-default}] pytest rc=$?"; tail -1 $out/pytest_$i.log | cut -c1-200
  done
fi
for rep in $(seq 1 $reps); do
  for i in "${!ARMS[@]}"; do
    env ${ARMS[$i]} timeout 1500 python bench.py --no-cpu-baseline --no-alt-engine $args --full-json $out/${i}_$rep.json > $out/${i}_$rep.line 2> $out/${i}_$rep.err
  done
done
python - "$out" "${ARMS[@]}" <<'PY'
import glob, json, os, sys
out, arms = sys.argv[1], sys.argv[2:]
def leaves(o, pre=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from leaves(v, pre + k + ".")
    elif isinstance(o, (int, float)) and not isinstance(o, bool):
        yield pre[:-1], o
keep = ("value", "ms_per_step", "roofline.frac", "roofline.whole_path_frac", "roofline.kernel_ms_per_step")
for f in sorted(glob.glob(os.path.join(out, "*_*.json"))):
    i = int(os.path.basename(f).split("_")[0])
    d = json.load(open(f))
    row = {k: v for k, v in leaves({k: d.get(k) for k in ("value", "ms_per_step", "roofline")}) if k in keep}
    for cn, c in (d.get("configs") or {}).items():
        if "error" in c:
            row[cn] = c["error"][:60]
            continue
        row[cn + ".ms"] = c.get("ms")
        for k, v in leaves(c.get("roofline", {})):
            if "frac" in k:
                row[cn + "." + k] = v
        for k, v in leaves(c.get("stage_ms", {})):
            row[cn + ".stage." + k] = v
    print("%-28s %s" % ("[%s] %s" % (arms[i] if i < len(arms) and arms[i] else "default", os.path.basename(f)),
                        "  ".join("%s=%s" % (k, ("%.5g" % v) if isinstance(v, float) else v) for k, v in row.items())))
PY
