#!/bin/bash
# The posterior's round-3 ladder on ONE box: every A/B switch of DESIGN 3.10 off, then on one by one (the subtracting
# epilogue's batched reads have no switch: they are in every line).
out=gpurun_out/${1:-ladder}
mkdir -p $out
Q="--no-cpu-baseline --no-alt-engine --no-parity-check --rows 1000000 --steps 1 --warmup 0 --configs posterior_f4096,posterior_f8257"
run() { tag=$1; shift; env "$@" python bench.py $Q > $out/$tag.json 2> $out/$tag.err; }
run 0_one_stream RR_POSDEF_OVERLAP=0 RR_CHOL_DIAG=0 RR_POSDEF_LOOKAHEAD=0 RR_GEMM64_K128=0 RR_SYRK64_TRI=0
run 1_substitution_on_second_stream RR_CHOL_DIAG=0 RR_POSDEF_LOOKAHEAD=0 RR_GEMM64_K128=0 RR_SYRK64_TRI=0
run 2_pipelined_diag_kernel RR_POSDEF_LOOKAHEAD=0 RR_GEMM64_K128=0 RR_SYRK64_TRI=0
run 3_lookahead RR_GEMM64_K128=0 RR_SYRK64_TRI=0
run 4_k128_products RR_SYRK64_TRI=0
run 5_triangular_syrk RR_NOTHING=1
python - <<PY
import json,glob
res={}
for f in sorted(glob.glob("$out/?_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1])["configs"]
        res[f.split("/")[-1][:-5]]={k:round(v["ms"],3) for k,v in d.items() if k.startswith("posterior")}
        print(f.split("/")[-1], res[f.split("/")[-1][:-5]])
json.dump({"command":"tools/r3_post_ladder.sh (one box, bench.py --configs posterior_f4096,posterior_f8257, median of 3 calls each)","ms":res}, open("$out/ladder.json","w"), indent=1)
PY
