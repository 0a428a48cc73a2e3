#!/bin/bash
# Round-5 trip 17: the k-major lean loop now also with a single-run K-contiguous partner (it was not eligible before:
# negative "jump" of a one-run operand) -- parity tests, then the in-place / K1-pass comparison again at every size.
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_workloads.py -m gpu -q --timeout 600 -k "view or k_major or mera or sliced or config or lean" > gpurun_out/r5_pytest_17.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5_pytest_17.log
timeout 300 python tools/kmajor_gate_probe2.py > gpurun_out/r5_kmajor_gate_small_lean.jsonl 2> gpurun_out/r5_kmajor_gate_small_lean.err; echo rc=$?; cat gpurun_out/r5_kmajor_gate_small_lean.jsonl
timeout 400 python tools/kmajor_gate_probe.py > gpurun_out/r5_kmajor_gate_lean.jsonl 2> gpurun_out/r5_kmajor_gate_lean.err; echo rc=$?; cat gpurun_out/r5_kmajor_gate_lean.jsonl
