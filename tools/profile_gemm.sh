#!/bin/bash
# PMC passes over one GEMM shape: VARIANT=... FILL=... bash tools/profile_gemm.sh
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pg_${TAG:-x}
mkdir -p $OUT
CMD="python $PWD/tools/gemm_one.py --variant ${VARIANT:-auto} --m 8192 --n 8192 --k ${K:-65536} --iters 3 --fill ${FILL:-uniform}"
cd /tmp
$CMD > $OUT/plain.json 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o g -- $CMD > $OUT/stats.log 2>&1
for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $ctr -d $OUT/prof_pmc_$tag -o g -- $CMD > $OUT/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
cat $OUT/plain.json
