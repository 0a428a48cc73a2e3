#!/bin/bash
# rocprofv3 evidence for the headline bench (run on the GPU box; results -> gpurun_out/prof_*).
# Pass 1: kernel trace + stats.  Passes 2..: PMC counters, each in its own run (no trace domains mixed in).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
BENCH="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-extras --no-verify --mera-chi 0 --svd-n 0 --rr-bond 0 ${BENCH_ARGS:-}"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o bench -- $BENCH > $OUT/prof_stats.log 2>&1
echo "stats rc=$?"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr -d $OUT/prof_pmc_$tag -o bench -- $BENCH > $OUT/prof_pmc_$tag.log 2>&1
  echo "pmc $ctr rc=$?"
done
find $OUT -name "*.csv" | head -40
