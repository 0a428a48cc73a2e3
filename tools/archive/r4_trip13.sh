#!/bin/bash
# Round 4, trip 13: mid-K split-K for f32 / f64 products with few tiles: GEMM tests, the MPS chain, DMRG sweeps.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mps.py tests/test_gpu_linalg.py -q --timeout 600 -k "gemm or tensordot or matmul or mps or dmrg or krylov or qr or eigh" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python tests/perf_mps_chain.py --d 2,4 2>&1 | tail -2 | cut -c1-220
timeout 300 python tests/perf_dmrg.py --bonds 256,512 --dtype float32 --cpu-max 0 2>&1 | tail -2
timeout 300 python tests/perf_dmrg.py --bonds 256 --dtype float64 --cpu-max 0 2>&1 | tail -1
