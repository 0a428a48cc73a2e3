#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t4; mkdir -p $O
timeout 300 python tools/svd_c128_diag2.py > $O/diag2.txt 2>&1; echo "rc=$?"; cat $O/diag2.txt
timeout 900 python -m pytest tests/test_gpu_svd_band.py -q --timeout 600 > $O/pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -8 $O/pytest_svd.log
