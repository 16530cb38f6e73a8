# Round-3 profiling recipe, run on the GPU box by gpurun (scratch under gpurun_out/prof_r03; tools/summarize_r03.py
# copies the summaries to profiles/r03_*).  Kernel traces and PMC passes are separate rocprofv3 runs.
#   sh tools/prof_r03.sh [stage ...]     stages: headline overlap elbo posdef predict laplace det c4 c5
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
mkdir -p $OUT
STAGES="${*:-headline overlap elbo posdef predict laplace}"
Q="--no-cpu-baseline --no-alt-engine --no-parity-check --config-timeout 400"
kt() {  # kt <tag> <bench args...>: kernel trace + stats
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o kt -- python bench.py $Q "$@" > $OUT/$tag.json 2> $OUT/$tag.err
}
pmc() {  # pmc <tag> "<counters>" <bench args...>
  tag=$1; ctr=$2; shift; shift
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$tag -o p -- python bench.py $Q "$@" > $OUT/$tag.json 2> $OUT/$tag.err
}
for st in $STAGES; do case $st in
headline)
  # the default command's headline part: N = 10M, 5 launches of 2M rows per pass, chunk k+1's features under chunk k's SYRK
  kt headline_kt --steps 3 --warmup 1 --configs none ;;
overlap)
  # A/B of the second-stream feature pass, un-profiled (the bench line's own clock), same box, alternating
  for rep in 1 2; do
    RR_GRAM_OVERLAP=0 python bench.py $Q --steps 3 --warmup 1 --configs none > $OUT/overlap_off_$rep.json 2> $OUT/overlap_off_$rep.err
    RR_GRAM_OVERLAP=1 python bench.py $Q --steps 3 --warmup 1 --configs none > $OUT/overlap_on_$rep.json 2> $OUT/overlap_on_$rep.err
  done
  RR_GRAM_OVERLAP=0 kt overlap_off_kt --steps 2 --warmup 1 --configs none ;;
elbo)
  kt elbo_kt --rows 1000000 --steps 1 --warmup 0 --configs c2_elbo_eval
  kt elbo64_kt --rows 1000000 --steps 1 --warmup 0 --configs c2f64_elbo_eval_n200k ;;
posdef)
  kt posdef_kt --rows 1000000 --steps 1 --warmup 0 --configs posterior_f4096,posterior_f8257 ;;
predict)
  kt predict_kt --rows 1000000 --steps 1 --warmup 0 --configs predict_moments_n300k ;;
laplace)
  kt laplace_kt --rows 1000000 --steps 1 --warmup 0 --configs c2laplace_f64phase_n1m ;;
det)
  RR_DETERMINISTIC=1 kt det_kt --steps 2 --warmup 1 --configs none
  RR_DETERMINISTIC=1 python bench.py $Q --steps 3 --warmup 1 --configs c2_elbo_eval > $OUT/det_bench.json 2> $OUT/det_bench.err ;;
c4)
  kt c4_kt --rows 1000000 --steps 1 --warmup 0 --configs c4
  pmc c4_fetch FETCH_SIZE --rows 1000000 --steps 1 --warmup 0 --configs c4
  pmc c4_write WRITE_SIZE --rows 1000000 --steps 1 --warmup 0 --configs c4 ;;
c5)
  kt c5_kt --rows 1000000 --steps 1 --warmup 0 --configs c5 ;;
c3)
  kt c3_kt --rows 1000000 --steps 1 --warmup 0 --configs c3 ;;
esac; done
find $OUT -name "*.csv" | wc -l
du -sh $OUT
# (appended) clock / pipe-busy counters of the GEMM kernel in the GLM step vs the SLM second pass:  sh tools/prof_r03.sh gemmclk
case " $* " in *" gemmclk "*)
  pmc c5_sq "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" --rows 1000000 --steps 1 --warmup 0 --configs c5
  pmc elbo_sq "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" --rows 1000000 --steps 1 --warmup 0 --configs c2_elbo_eval ;;
esac
# (appended) L2->fabric traffic of the headline kernels, separate PMC passes:  sh tools/prof_r03.sh hbm
case " $* " in *" hbm "*)
  pmc headline_fetch FETCH_SIZE --rows 2097152 --steps 1 --warmup 0 --configs none
  pmc headline_write WRITE_SIZE --rows 2097152 --steps 1 --warmup 0 --configs none ;;
esac
# (appended) the same with the default command's own chunking (N = 10M), one pass:  sh tools/prof_r03.sh hbm10m
case " $* " in *" hbm10m "*)
  pmc headline10m_fetch FETCH_SIZE --steps 1 --warmup 0 --configs none
  for rep in 1 2; do
    RR_GRAM_CHUNK_ROWS=2097152 python bench.py $Q --steps 3 --warmup 1 --configs none > $OUT/chunk_pow2_$rep.json 2> $OUT/chunk_pow2_$rep.err
    python bench.py $Q --steps 3 --warmup 1 --configs none > $OUT/chunk_equal_$rep.json 2> $OUT/chunk_equal_$rep.err
  done ;;
esac
# (appended) matrix-pipe busy cycles and clock of the headline kernels, one 2M-row launch each:  sh tools/prof_r03.sh sq
case " $* " in *" sq "*)
  pmc headline_sq "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" --rows 2000000 --steps 1 --warmup 0 --configs none
  pmc elbo_sq2 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" --rows 1000000 --steps 1 --warmup 0 --configs c2_elbo_eval ;;
esac
