#!/bin/bash
# Round-5 trip 4: (1) the headline shape through the plain NT entry point vs the view kernel (same buffers);
# (2) one chi = 64 MERA placement with the HBM-sized stage cache (executed multiply-adds must equal the model's).
set -u
mkdir -p gpurun_out
timeout 300 python tools/gemm_r5_probe.py --parity 0 --shapes 8192x8192x8192 --fills normal --variants auto \
  --headline_variants auto,plain:auto,auto:r0,plain:auto:r0,auto,plain:auto > gpurun_out/r5_probe4.jsonl 2> gpurun_out/r5_probe4.err; echo "probe rc=$?"; tail -2 gpurun_out/r5_probe4.err
timeout 400 python - > gpurun_out/r5_mera64_cache.json 2> gpurun_out/r5_mera64_cache.err <<'PY'
import json, time
import tensornetwork_amd as ta
from tensornetwork_amd import workloads
be = ta.get_hip_backend()
run = workloads.mera_sliced_run(be, 64, "left", ta.bfloat16, budget_seconds=200.0, check_every=1024)
run.pop("semantics", None)
print(json.dumps(run))
PY
echo "mera rc=$?"; tail -2 gpurun_out/r5_mera64_cache.err; head -c 1500 gpurun_out/r5_mera64_cache.json
