#!/bin/bash
# Round 6, trip 7: index_update with a tensor assignee on the GPU; the Sturm kernel choice for small rounds.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "index_update or tiny or signature or high_rank" > $OUT/t7_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/t7_pytest.log
timeout 300 python tools/svd_fast_probe.py --sizes 512x512,768x768,1024x1024,1000x600,4096x512,4096x4096 --spectra 0 > $OUT/t7_fast_probe.jsonl 2> $OUT/t7_fast_probe.err; echo "probe rc=$?"
cut -c1-200 $OUT/t7_fast_probe.jsonl; tail -3 $OUT/t7_fast_probe.err
echo "== TNH_SVDB_LANE_AUTO=0"; TNH_SVDB_LANE_AUTO=0 timeout 300 python tools/svd_fast_probe.py --sizes 512x512,768x768,1024x1024,1000x600 --spectra 0 --check 0 2>&1 | grep '"fast_env": 1' | cut -c1-160
timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py tests/test_gpu_mps.py -m gpu -q --timeout 900 > $OUT/t7_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -4 $OUT/t7_pytest_svd.log
