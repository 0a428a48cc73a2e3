#!/bin/bash
# Round 6, trip 9: tiny kernel with loads in flight; the d = 4 chain's shapes.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "tiny or split_k" > $OUT/t9_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/t9_pytest.log
timeout 300 python tests/perf_mps_chain.py --D 512 --d 2,4 > $OUT/t9_mps.log 2>&1; echo "mps rc=$?"; tail -3 $OUT/t9_mps.log | cut -c1-300
timeout 300 python tools/mps_chain_shapes.py --d 4 > $OUT/t9_mps_shapes_d4.jsonl 2>&1; tail -48 $OUT/t9_mps_shapes_d4.jsonl | cut -c1-160
