#!/bin/bash
# Round 6, trip 12: kernel table of the f64 band SVD with the fast stage.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/prof_svd_f32 $OUT/prof_svd_f64
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f64 -o svd -- python $OUT/../tools/svd_stats_run.py f64 > $OUT/t12_svd_f64.log 2>&1; echo "svd prof rc=$?")
find $OUT/prof_svd_f64 -name "*kernel_trace.csv" -delete
python tools/svd_stats_summary.py $OUT $OUT | tail -30
