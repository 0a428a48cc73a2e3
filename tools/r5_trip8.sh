#!/bin/bash
# Round-5 trip 8: lean loop for every K-contiguous view form (half-K-tile walk included) + scalar loop control.
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_workloads.py -m gpu -q --timeout 600 -k "gemm or view or mera or sliced or config" > gpurun_out/r5_pytest_gemm4.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r5_pytest_gemm3.log
timeout 300 python tools/gemm_r5_probe.py --parity 0 --shapes 8192x8192x8192 --fills normal --variants auto,auto:l0 \
  --headline_variants auto,auto:l0,auto > gpurun_out/r5_probe9.jsonl 2> gpurun_out/r5_probe8.err; echo "probe rc=$?"; tail -2 gpurun_out/r5_probe8.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mera-chi 32 --svd-n 0 --rr-bond 16 --rr-bond-small 12 > gpurun_out/r5_bench9.json 2> gpurun_out/r5_bench8.err; echo "bench rc=$?"; tail -c 2000 gpurun_out/r5_bench8.json
