#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python tools/svd_sizes_probe.py > $O/svd_sizes.jsonl 2> $O/svd_sizes.err; cut -c1-120 $O/svd_sizes.jsonl
