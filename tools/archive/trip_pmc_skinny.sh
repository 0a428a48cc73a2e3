#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
CMD="python $R/tools/gemm_one.py --variant auto --m 144 --n 2985984 --k 144 --iters 5 --fill uniform"
cd /tmp
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-30)
  rm -rf $OUT/prof_pmc_sk_$tag
  timeout 200 rocprofv3 --pmc $ctr -d $OUT/prof_pmc_sk_$tag -o sk -- $CMD > $OUT/pmc_sk_$tag.log 2>&1
  echo "pmc $ctr rc=$?"
done
cd $R
python - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob('gpurun_out/prof_pmc_sk_*/*.db')):
  c = sqlite3.connect(db)
  q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%ragged%' group by kernel_name, counter_name")
  for name, ctr, n, avg in c.execute(q):
    print(f"{ctr:28s} n={n:3d} avg={avg:18.1f}")
PY
