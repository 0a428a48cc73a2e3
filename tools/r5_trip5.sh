#!/bin/bash
# Round-5 trip 5: the K-walk forms of the view kernel -- parity tests, then the headline shape per form.
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "view" > gpurun_out/r5_pytest_view.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5_pytest_view.log
timeout 300 python tools/gemm_r5_probe.py --parity 0 --shapes 8192x8192x8192 --fills normal --variants auto \
  --headline_variants auto,auto:w0,auto:w1,plain:auto,auto,auto:w0 > gpurun_out/r5_probe5.jsonl 2> gpurun_out/r5_probe5.err; echo "probe rc=$?"; tail -2 gpurun_out/r5_probe5.err
