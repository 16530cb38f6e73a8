# Profiling recipe run on the GPU box by gpurun (outputs under gpurun_out/prof, summaries are then
# copied to profiles/).  Kernel trace and each PMC pass are separate rocprofv3 runs.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
# ENGINE=bf16x3 sh tools/prof.sh profiles the split-bf16 engine instead (summarise with: summarize_prof.py <tag> rr_syrk_b16w4_kernel)
EXTRA="--no-alt-engine --no-parity-check ${ENGINE:+--engine $ENGINE}"
CMD="python bench.py --rows 2000000 --steps 3 --warmup 1 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt_bench.json 2> $OUT/kt.err
PMC="python bench.py --rows 2000000 --steps 1 --warmup 0 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $PMC > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $PMC > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- $PMC > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_lds -o p -- $PMC > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err
find $OUT -name "*.csv" | head -40
du -sh $OUT
