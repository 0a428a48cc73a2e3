#!/bin/bash
# Round 6, trip 5: reduce kernels with batched partial loads; skinny split-K on the MPS chain.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/svd_fast_probe.py --sizes 4096x4096,2048x2048,1024x1024,512x512 --spectra 0 > $OUT/t5_fast_probe.jsonl 2> $OUT/t5_fast_probe.err; echo "probe rc=$?"
cut -c1-200 $OUT/t5_fast_probe.jsonl; tail -3 $OUT/t5_fast_probe.err
timeout 900 python -m pytest tests/test_gpu_svd_band.py -m gpu -q --timeout 900 -x > $OUT/t5_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -4 $OUT/t5_pytest_svd.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -k "split_k" > $OUT/t5_pytest_splitk.log 2>&1; echo "pytest splitk rc=$?"; tail -8 $OUT/t5_pytest_splitk.log
rm -rf $OUT/prof_svd_f32
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t5_svd_f32.log 2>&1; echo "svd prof rc=$?")
find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
python tools/svd_stats_summary.py $OUT $OUT | tail -22
timeout 300 python tests/perf_mps_chain.py --D 512 --d 2,4 > $OUT/t5_mps.log 2>&1; echo "mps rc=$?"; tail -3 $OUT/t5_mps.log | cut -c1-300
timeout 300 python tools/mps_chain_shapes.py > $OUT/t5_mps_shapes.jsonl 2>&1; tail -12 $OUT/t5_mps_shapes.jsonl | cut -c1-200
