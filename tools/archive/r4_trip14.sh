#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t14; mkdir -p $O
timeout 240 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_graph.py tests/test_gpu_mps.py -m gpu -q --timeout 200 --durations=8 > $O/pytest.log 2>&1; echo "rc=$?"; tail -16 $O/pytest.log
