#!/bin/bash
# Round-5 trip 16: PMC counters of the k-major (nn) view kernel against the NT view kernel at D = 128.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
for mode in inplace permute; do
  for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo ${mode}_$ctr | tr ' ' '_' | cut -c1-40)
    rm -rf $OUT/prof_km_$tag
    (cd /tmp && timeout 200 rocprofv3 --pmc $ctr -d $OUT/prof_km_$tag -o km -- python $OUT/../tools/kmajor_pmc_run.py $mode > $OUT/prof_km_$tag.log 2>&1; echo "$tag rc=$?")
  done
done
python - <<'PY'
import glob, os, sqlite3
out = open("gpurun_out/r5_kmajor_pmc.txt", "w")
for db in sorted(glob.glob("gpurun_out/prof_km_*/**/*.db", recursive=True)):
  c = sqlite3.connect(db)
  out.write(f"# {os.path.relpath(db, 'gpurun_out')}\n")
  q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%gemm_nt_pp%' "
       "group by kernel_name, counter_name order by kernel_name, counter_name")
  try:
    for name, ctr, n, avg in c.execute(q):
      out.write(f"{ctr:28s} n={n:3d} avg={avg:18.1f}  {name[:100]}\n")
  except sqlite3.Error as e:
    out.write(f"error {e}\n")
out.close()
print(open("gpurun_out/r5_kmajor_pmc.txt").read())
PY
rm -rf $OUT/prof_km_*
