#!/bin/bash
# Round 6, trip 2: the fast band reduction -- band SVD tests, the A/B probe, a kernel table.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/svd_fast_probe.py > $OUT/t2_fast_probe.jsonl 2> $OUT/t2_fast_probe.err; echo "probe rc=$?"
cat $OUT/t2_fast_probe.jsonl | cut -c1-400; tail -5 $OUT/t2_fast_probe.err
timeout 1500 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -m gpu -q --timeout 900 > $OUT/t2_pytest_svd.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/t2_pytest_svd.log
rm -rf $OUT/prof_svd_f32
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t2_svd_f32.log 2>&1; echo "svd prof rc=$?")
find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
python tools/svd_stats_summary.py $OUT $OUT
cat $OUT/svd_band_f32_kernel_stats.txt
