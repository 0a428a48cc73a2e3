#!/bin/bash
# Round 6, trip 4: reduce kernels shared by four waves (ds_read_b128 operands).
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/svd_fast_probe.py --sizes 4096x4096,2048x2048,1024x1024,512x512,3072x1024,1000x600 --spectra 1 > $OUT/t4_fast_probe.jsonl 2> $OUT/t4_fast_probe.err; echo "probe rc=$?"
cut -c1-230 $OUT/t4_fast_probe.jsonl; tail -3 $OUT/t4_fast_probe.err
timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -m gpu -q --timeout 900 > $OUT/t4_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -4 $OUT/t4_pytest_svd.log
rm -rf $OUT/prof_svd_f32
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t4_svd_f32.log 2>&1; echo "svd prof rc=$?")
find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
python tools/svd_stats_summary.py $OUT $OUT | tail -22
