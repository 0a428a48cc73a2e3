#!/bin/bash
# Round 6, trip 11: the fast band reduction for f64 input.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/svd_fast_probe.py --dtype f64 --sizes 1024x1024,2048x2048,4096x4096,512x512,3072x1024,1000x600 --spectra 1 --reps 3 > $OUT/t11_fast_probe_f64.jsonl 2> $OUT/t11_fast_probe_f64.err; echo "probe rc=$?"
cut -c1-330 $OUT/t11_fast_probe_f64.jsonl; tail -5 $OUT/t11_fast_probe_f64.err
timeout 300 python tools/svd_fast_probe.py --sizes 4096x4096,1024x1024 --spectra 0 --reps 3 2>&1 | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py tests/test_gpu_mps.py -m gpu -q --timeout 900 > $OUT/t11_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -15 $OUT/t11_pytest_svd.log
