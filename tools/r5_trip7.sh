#!/bin/bash
# Round-5 trip 7: lean loop generalised (two-level rows, tile-granular K walk), M0 writes interleaved with fragment reads.
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "gemm or view" > gpurun_out/r5_pytest_gemm2.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r5_pytest_gemm2.log
timeout 400 python tools/gemm_r5_probe.py --parity 0 --shapes 8192x8192x8192 --fills normal,zeros --variants auto,auto:l0 \
  --headline_variants auto,auto:l0,auto,auto:l0 > gpurun_out/r5_probe7.jsonl 2> gpurun_out/r5_probe7.err; echo "probe rc=$?"; tail -2 gpurun_out/r5_probe7.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mera-chi 32 --svd-n 0 --rr-bond 0 > gpurun_out/r5_bench7.json 2> gpurun_out/r5_bench7.err; echo "bench rc=$?"; tail -c 1800 gpurun_out/r5_bench7.json
