#!/bin/bash
# Round 6, trip 1: GEMM kernel tests under the tightened bf16 / f16 rule + the BASELINE-size tests, the band SVD's
# kernel table (baseline of the round), a kernel trace of the configs[3] MPS chain under graph replay.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -x -k "gemm or config2 or d512" > $OUT/t1_pytest_gemm.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/t1_pytest_gemm.log
rm -rf $OUT/prof_svd_f32 $OUT/prof_mps
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t1_svd_f32.log 2>&1; echo "svd prof rc=$?")
find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
python tools/svd_stats_summary.py $OUT $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_mps -o mps -- python $OUT/../tests/perf_mps_chain.py --D 512 --d 2 > $OUT/t1_mps.log 2>&1; echo "mps prof rc=$?")
cat $OUT/t1_mps.log | tail -3
python - <<'PY'
import glob, sqlite3, os
for db in glob.glob(os.path.join(os.environ.get("OUT", "gpurun_out"), "prof_mps", "*.db")) + glob.glob("gpurun_out/prof_mps/**/*.db", recursive=True):
  rows = list(sqlite3.connect(db).execute("select * from top_kernels"))
  for r in rows[:25]: print(r)
  break
PY
find $OUT/prof_mps -name "*kernel_trace.csv" | head -2
