#!/bin/bash
# Round 6, trip 8: the skinny f32 / f64 kernel; MPS chain; the whole kernel test file.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -k "skinny or split_k or tiny" > $OUT/t8_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/t8_pytest.log
timeout 300 python tests/perf_mps_chain.py --D 512 --d 2,4 > $OUT/t8_mps.log 2>&1; echo "mps rc=$?"; tail -3 $OUT/t8_mps.log | cut -c1-300
timeout 300 python tools/mps_chain_shapes.py > $OUT/t8_mps_shapes.jsonl 2>&1; tail -48 $OUT/t8_mps_shapes.jsonl | cut -c1-160
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mps.py tests/test_gpu_workloads.py tests/test_gpu_graph.py -m gpu -q --timeout 900 > $OUT/t8_pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -12 $OUT/t8_pytest_all.log
