#!/bin/bash
# Gather lowering ON by default: the GPU tests whose products it can touch, then bench.py's sliced-network leg alone.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t21; mkdir -p $O
timeout 170 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tensordot or gather or gemm or stream or matmul" --timeout 160 > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?"; tail -3 $O/pytest_kernels.log
timeout 120 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_graph.py -m gpu -q -x -k "not rccl" --timeout 110 > $O/pytest_workloads.log 2>&1; echo "pytest workloads rc=$?"; tail -3 $O/pytest_workloads.log
timeout 100 python tools/rr64_check.py > $O/rr64_check.json 2> $O/rr64_check.err; echo "check rc=$?"; cat $O/rr64_check.json; tail -3 $O/rr64_check.err
