#!/bin/bash
# Round-5 trip 15: the k-major cost rule and the strided in-place read on the workloads (MERA chi = 32, sliced networks, sweep).
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_workloads.py -m gpu -q --timeout 600 -k "view or k_major or mera or sliced or config or lean" > gpurun_out/r5_pytest_15.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5_pytest_15.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mera-chi 32 --svd-n 0 --rr-bond 16 --rr-bond-small 12 > gpurun_out/r5_bench15.json 2> gpurun_out/r5_bench15.err; echo "bench rc=$?"; tail -c 2000 gpurun_out/r5_bench15.json
timeout 300 python - > gpurun_out/r5_bench15_nopenalty.json 2> gpurun_out/r5_bench15_nopenalty.err <<'PY'
import json, sys
import tensornetwork_amd as ta
from tensornetwork_amd import hip_backend
hip_backend.HipBackend.kmajor_inplace_penalty = 0.0      # the rule of rounds 2-4, for comparison on the same box
import bench
be = ta.get_hip_backend()
print(json.dumps({"mera_chi32_without_rule": bench.mera_bench(ta, be, 32, verify=False)}))
print(json.dumps({"sliced_D16_without_rule": bench.sliced_network_bench(ta, be, None, 0, 1, 16, 64, False)}))
PY
echo "nopenalty rc=$?"; cut -c1-600 gpurun_out/r5_bench15_nopenalty.json
