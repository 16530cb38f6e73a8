# Round-2 profiling recipe, run on the GPU box by gpurun (scratch under gpurun_out/prof_r02; tools/summarize_r02.py
# copies the summaries to profiles/r02_*).  Kernel trace and every PMC pass are separate rocprofv3 runs.
#   sh tools/prof_r02.sh [stage ...]     stages: counters default headline configs hbm mall engine
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02
mkdir -p $OUT
STAGES="${*:-counters headline configs hbm}"
Q="--no-cpu-baseline --no-alt-engine --no-parity-check"
kt() {  # kt <tag> <bench args...>: kernel trace + stats
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o kt -- python bench.py $Q "$@" > $OUT/$tag.json 2> $OUT/$tag.err
}
pmc() {  # pmc <tag> "<counters>" <bench args...>
  tag=$1; ctr=$2; shift; shift
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$tag -o p -- python bench.py $Q "$@" > $OUT/$tag.json 2> $OUT/$tag.err
}
for st in $STAGES; do case $st in
counters)
  rocprofv3 -L > $OUT/counters_all.txt 2>&1
  grep -i -E "dram|mall|hbm|EA0?_RD|EA0?_WR|FETCH_SIZE|WRITE_SIZE" $OUT/counters_all.txt | head -80 > $OUT/counters_mem.txt ;;
headline)
  kt headline_kt --rows 2000000 --steps 3 --warmup 1 --configs none
  pmc headline_fetch FETCH_SIZE --rows 2000000 --steps 1 --warmup 0 --configs none
  pmc headline_write WRITE_SIZE --rows 2000000 --steps 1 --warmup 0 --configs none
  pmc headline_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" --rows 2000000 --steps 1 --warmup 0 --configs none ;;
default)
  # the default bench command's headline part (N = 10M: 4 passes of 5 launches of 2M rows), without the CPU baseline, the
  # 2048-row parity launch and the other configurations, which would mix other launch sizes into the kernel averages
  kt default_kt --configs none ;;
configs)
  for cfg in headline c3 c4 c5; do
    kt cfg_${cfg}_kt --rows 500000 --steps 1 --warmup 0 --configs $cfg
  done ;;
hbm)
  pmc cfg_c4_fetch FETCH_SIZE --rows 500000 --steps 1 --warmup 0 --configs c4
  pmc cfg_c4_write WRITE_SIZE --rows 500000 --steps 1 --warmup 0 --configs c4 ;;
mall)
  # does the SYRK's L2->fabric fetch come from HBM or from the Infinity Cache?  One launch over 16 384 rows keeps its
  # 256 MiB of P inside the 256 MiB MALL; the regular 2M-row launch streams 32.8 GB.  $MALL_CTRS: DRAM-side counters found
  # by the `counters` stage.
  for rows in 16384 2000000; do
    RR_GRAM_CHUNK_ROWS=$rows pmc mall_${rows}_fetch FETCH_SIZE --rows 2000000 --steps 1 --warmup 0 --configs none
    [ -n "$MALL_CTRS" ] && RR_GRAM_CHUNK_ROWS=$rows pmc mall_${rows}_dram "$MALL_CTRS" --rows 2000000 --steps 1 --warmup 0 --configs none
  done ;;
engine)
  kt fp16x3_kt --rows 2000000 --steps 3 --warmup 1 --configs none --engine fp16x3
  pmc fp16x3_fetch FETCH_SIZE --rows 2000000 --steps 1 --warmup 0 --configs none --engine fp16x3
  pmc fp16x3_write WRITE_SIZE --rows 2000000 --steps 1 --warmup 0 --configs none --engine fp16x3
  pmc fp16x3_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" --rows 2000000 --steps 1 --warmup 0 --configs none --engine fp16x3 ;;
esac; done
find $OUT -name "*.csv" | wc -l
du -sh $OUT
