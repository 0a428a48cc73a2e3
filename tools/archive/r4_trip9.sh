#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_mps.py -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for dt in float32 float64; do
  timeout 400 python tests/perf_dmrg.py --bonds 256,512 --dtype $dt --cpu-max 0 2>&1 | tail -2 | tee -a $O/dmrg.txt
done
