#!/bin/bash
# Round-5 trip 18: PMC counters of the 32x32x16 variant (:p3) against the lean default at 8192^3, zero-filled and random.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
for v in "bf16_256pp:p3" "bf16_256pp" "bf16_256pp:l0"; do
  for fill in zeros uniform; do
    for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
      tag=$(echo ${v}_${fill}_$ctr | tr ' :' '__' | cut -c1-48)
      rm -rf $OUT/prof_m32_$tag
      (cd /tmp && timeout 120 rocprofv3 --pmc $ctr -d $OUT/prof_m32_$tag -o m32 -- python $OUT/../tools/gemm_one.py --variant $v --m 8192 --n 8192 --k 8192 --iters 4 --fill $fill > $OUT/prof_m32_$tag.log 2>&1; echo "$tag rc=$?")
    done
  done
done
python - <<'PY'
import glob, os, sqlite3
out = open("gpurun_out/r5_m32_pmc.txt", "w")
for db in sorted(glob.glob("gpurun_out/prof_m32_*/**/*.db", recursive=True)):
  c = sqlite3.connect(db)
  out.write(f"# {os.path.relpath(db, 'gpurun_out')}\n")
  q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%gemm_nt_pp%' "
       "group by kernel_name, counter_name order by kernel_name, counter_name")
  try:
    for name, ctr, n, avg in c.execute(q):
      out.write(f"{ctr:28s} n={n:3d} avg={avg:18.1f}  {name[45:110]}\n")
  except sqlite3.Error as e:
    out.write(f"error {e}\n")
out.close()
print(open("gpurun_out/r5_m32_pmc.txt").read())
PY
rm -rf $OUT/prof_m32_*
