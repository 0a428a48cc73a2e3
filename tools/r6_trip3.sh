#!/bin/bash
# Round 6, trip 3: fast stage 1 with the scalar-operand reduce kernels; tile knobs; new boundary tests; MPS chain shapes.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/svd_fast_probe.py --sizes 4096x4096,2048x2048,1024x1024,512x512 --spectra 1 > $OUT/t3_fast_probe.jsonl 2> $OUT/t3_fast_probe.err; echo "probe rc=$?"
cut -c1-230 $OUT/t3_fast_probe.jsonl; tail -3 $OUT/t3_fast_probe.err
for knob in "TNH_SVDB_FAST_CW=64" "TNH_SVDB_FAST_CW=128" "TNH_SVDB_FAST_WGS=512" "TNH_SVDB_FAST_WGS=2048" "TNH_SVDB_FAST_SWITCH=64" "TNH_SVDB_FAST_SWITCH=256"; do
  echo "== $knob"; env $knob timeout 200 python tools/svd_fast_probe.py --sizes 4096x4096,2048x2048 --spectra 0 --check 0 2>&1 | grep '"fast_env": 1' | cut -c1-200
done | tee $OUT/t3_knobs.txt
timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -m gpu -q --timeout 900 > $OUT/t3_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -4 $OUT/t3_pytest_svd.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "signature or high_rank" > $OUT/t3_pytest_boundary.log 2>&1; echo "pytest boundary rc=$?"; tail -8 $OUT/t3_pytest_boundary.log
rm -rf $OUT/prof_svd_f32
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t3_svd_f32.log 2>&1; echo "svd prof rc=$?")
find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
python tools/svd_stats_summary.py $OUT $OUT | tail -25
timeout 300 python tools/mps_chain_shapes.py > $OUT/t3_mps_shapes.jsonl 2>&1; echo "mps shapes rc=$?"; cat $OUT/t3_mps_shapes.jsonl | cut -c1-200
