#!/bin/bash
# Round 6, trip 6: tiny-output kernel, host overhead of small steps.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -k "split_k or tiny" > $OUT/t6_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/t6_pytest.log
timeout 300 python tests/perf_mps_chain.py --D 512 --d 2,4 > $OUT/t6_mps.log 2>&1; echo "mps rc=$?"; tail -3 $OUT/t6_mps.log | cut -c1-300
timeout 600 python tools/host_overhead_probe.py > $OUT/t6_host.txt 2>&1; echo "host rc=$?"; head -70 $OUT/t6_host.txt | cut -c1-200; grep -n "mps chain" -A60 $OUT/t6_host.txt | cut -c1-200
