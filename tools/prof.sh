# Profiling recipe, run on the GPU box by gpurun; scratch under gpurun_out/prof_$ROUND (default r05), summarised into
# profiles/${ROUND}_* by tools/summarize.py.  Kernel traces and PMC passes are SEPARATE rocprofv3 runs (gpurun refuses
# --pmc together with the hip / hsa / memory-copy trace domains).
#   sh tools/prof.sh [stage ...]
# stages (kernel trace + stats of `python bench.py --configs <that configuration>`):
#   headline  elbo  elbo64  c3  ffelbo  posdef  predict  laplace  c4  c5  c1  c4gm  sp (bench.py --gpus 2 --single-process)
# counter passes:
#   sq     matrix-pipe busy cycles + clock of the headline kernels and of the two second-pass kernels (predictsq: predict_moments' product;
#          c5sq: the GLM step's three products)
#   hbm    FETCH_SIZE / WRITE_SIZE of the headline command's launches (profiles/traffic.json)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROUND=${ROUND:-r06}
OUT=gpurun_out/prof_$ROUND
mkdir -p $OUT
STAGES="${*:-headline elbo c3 ffelbo}"
Q="--no-cpu-baseline --no-alt-engine --no-parity-check --config-timeout 600"
S="--rows 1000000 --steps 1 --warmup 0"
kt() {  # kt <tag> <bench args...>: kernel trace + stats; the bench's unabridged record next to it
  tag=$1; shift
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o kt -- python bench.py $Q --full-json $OUT/$tag.full.json "$@" > $OUT/$tag.json 2> $OUT/$tag.err
}
pmc() {  # pmc <tag> "<counters>" <bench args...>
  tag=$1; ctr=$2; shift; shift
  timeout 1200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$tag -o p -- python bench.py $Q --full-json $OUT/$tag.full.json "$@" > $OUT/$tag.json 2> $OUT/$tag.err
}
SQ="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"
for st in $STAGES; do case $st in
headline) kt headline_kt --steps 3 --warmup 1 --configs none ;;
elbo)     kt elbo_kt $S --configs c2_elbo_eval ;;
elbo64)   kt elbo64_kt $S --configs c2f64_elbo_eval_n200k ;;
c3)       kt c3_kt $S --configs c3 ;;
ffelbo)   kt ffelbo_kt $S --configs c4elbo ;;
posdef)   kt posdef_kt $S --configs posterior_f4096,posterior_f8257,posterior_f16384 ;;
predict)  kt predict_kt $S --configs predict_moments_n300k ;;
laplace)  kt laplace_kt $S --configs c2laplace_f64phase_n1m ;;
c4)       kt c4_kt $S --configs c4 ;;
c5)       kt c5_kt $S --configs c5 ;;
c1)       kt c1_kt $S --configs c1 ;;
c4gm)     kt c4gm_kt $S --configs c4gm ;;
c5small)  kt c5small_kt $S --configs c5small ;;
sp)       timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sp_kt -o kt -- python bench.py --gpus 2 --single-process --steps 3 --warmup 1 --rows 4000000 --dist-rows 1000000 --no-parity-check --full-json $OUT/sp_kt.full.json > $OUT/sp_kt.json 2> $OUT/sp_kt.err ;;
sq)
  pmc headline_sq "$SQ" --rows 2000000 --steps 1 --warmup 0 --configs none
  pmc elbo_sq "$SQ" $S --configs c2_elbo_eval
  pmc c3_sq "$SQ" $S --configs c3 ;;
predictsq) pmc predict_sq "$SQ" $S --configs predict_moments_n300k ;;
c5sq) pmc c5_sq "$SQ" $S --configs c5 ;;
elbo64sq) pmc elbo64_sq "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" $S --configs c2f64_elbo_eval_n200k,headline_shape_f64 ;;
hbm)
  pmc headline_fetch FETCH_SIZE --steps 1 --warmup 0 --configs none
  pmc headline_write WRITE_SIZE --steps 1 --warmup 0 --configs none ;;
esac; done
find $OUT -name "*.csv" | wc -l
du -sh $OUT
