#!/bin/bash
# Round-5 trip 6: the lean main loop -- parity tests (GEMM file), then 8192^3 / long-K / headline with and without it.
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "gemm" > gpurun_out/r5_pytest_gemm.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r5_pytest_gemm.log
timeout 400 python tools/gemm_r5_probe.py --parity 0 --shapes 8192x8192x8192,8192x8192x65536 --fills normal,zeros --variants auto,auto:l0 \
  --headline_variants auto,auto:l0,plain:auto,plain:auto:l0,auto,auto:l0 > gpurun_out/r5_probe6.jsonl 2> gpurun_out/r5_probe6.err; echo "probe rc=$?"; tail -2 gpurun_out/r5_probe6.err
