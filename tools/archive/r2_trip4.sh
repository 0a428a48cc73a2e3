#!/bin/bash
# Round-2 trip 4: GPU suite after the policy / reduction / permute changes, SVD convergence probe, drop-in rerun.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TN_REFERENCE_DIR=$PWD/_reference_scratch
echo "== reference drop-in"
OUT=$OUT/refdropin PER_FILE_TIMEOUT=600 timeout 1500 bash tools/reference_dropin/run_reference_tests.sh 2>&1 | tail -16
echo "== pytest gpu (without the drop-in file: run above)"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_reference_dropin.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
echo "== svd probe"
timeout 600 python tools/svd_sweep_probe.py --n 4096 > $OUT/svd_sweep_4096.jsonl 2> $OUT/svd_sweep_4096.err; echo "rc=$?"; cat $OUT/svd_sweep_4096.jsonl; grep "tnh svd" $OUT/svd_sweep_4096.err | head -40
timeout 600 python tools/svd_sweep_probe.py --n 2048 --spectrum graded > $OUT/svd_sweep_2048g.jsonl 2> $OUT/svd_sweep_2048g.err; echo "rc=$?"; cat $OUT/svd_sweep_2048g.jsonl; grep "tnh svd" $OUT/svd_sweep_2048g.err | head -40
echo "== helpers"
timeout 600 python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench, tensornetwork_amd as ta
be = ta.get_hip_backend()
for row in bench.helpers_bench(ta, be):
  print(json.dumps(row))
PY
