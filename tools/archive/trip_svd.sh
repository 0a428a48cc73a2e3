#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_workloads.py tests/test_gpu_linalg.py tests/test_gpu_mps.py -m gpu -q -x --timeout 300 -k "svd or split or mps or dmrg or eigh or inv" > gpurun_out/pytest_svd.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_svd.log
timeout 300 python tools/svd_probe.py --check 0 --sizes 4096 --reps 1 2>&1 | tail -3
timeout 200 python bench.py --steps 1 --warmup 1 --rr-bond 0 --no-sweep --mera-chi 0 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
  if l.startswith('{'): print(json.loads(l).get('svd'))
"
