#!/bin/bash
set -u
O=gpurun_out/${1:-r3t32}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 600 -k "narrow or integer or int32 or int64 or dtype" > $O/pytest_int.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest_int.log
