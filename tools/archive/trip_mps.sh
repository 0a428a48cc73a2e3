#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mps.py -m gpu -q -x --timeout 600 > gpurun_out/pytest_mps.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/pytest_mps.log
timeout 900 python tests/perf_dmrg.py --n 24 --bonds 64,128 --cpu-max 128 2>&1 | tail -6
