#!/bin/bash
# Round-2 trip 20: two-group SVD schedule (two streams, sync at phase boundaries only): tests, timing A/B.
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_linalg.py -m gpu -q --timeout 600 > $OUT/pytest_svd.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_svd.log
echo "== A/B"
for sc in 4 2 1; do
  echo "sched=$sc: $(TNH_SVD_SCHED=$sc timeout 300 python tools/svd_probe.py --check 1 --sizes 4096,2048,1024,512 --reps 2 2>&1 | tail -5 | tr '\n' ' ')"
done
