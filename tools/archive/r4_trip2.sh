#!/bin/bash
# Round 4, trip 2: the templated band SVD (f32 unchanged? f64 new), padded sides, pad_results test, K8 bring-up on a thread.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -q --timeout 600 -x > $O/pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -25 $O/pytest_svd.log
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_workloads.py -q --timeout 300 -k "row_padded or k8 or rccl" > $O/pytest_misc.log 2>&1; echo "pytest misc rc=$?"; tail -5 $O/pytest_misc.log
timeout 600 python tools/svd_sizes_probe.py > $O/svd_sizes.jsonl 2> $O/svd_sizes.err; echo "probe rc=$?"; cat $O/svd_sizes.jsonl; tail -3 $O/svd_sizes.err
