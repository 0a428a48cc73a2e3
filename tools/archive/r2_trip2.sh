#!/bin/bash
# Round-2 trip 2: full GPU suite (ints, K8, view GEMM, reference drop-in), view/pad probe, power telemetry.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TN_REFERENCE_DIR=$PWD/_reference_scratch
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
echo "== view probe"
timeout 600 python tools/view_probe.py > $OUT/view_probe.jsonl 2> $OUT/view_probe.err; echo "rc=$?"; cat $OUT/view_probe.jsonl; tail -5 $OUT/view_probe.err
echo "== power probe"
timeout 400 python tools/power_probe.py --seconds 2.0 --shapes 8192x8192x8192,8192x8192x65536 > $OUT/power_probe.jsonl 2> $OUT/power_probe.err; echo "rc=$?"; cut -c1-600 $OUT/power_probe.jsonl; tail -3 $OUT/power_probe.err
echo "== reference drop-in"
OUT=$OUT/refdropin PER_FILE_TIMEOUT=600 timeout 1500 bash tools/reference_dropin/run_reference_tests.sh 2>&1 | tail -16
