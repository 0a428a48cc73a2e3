#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_linalg.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_linalg.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/pytest_linalg.log
timeout 300 python tools/qr_probe.py 2>&1 | tail -12
