#!/bin/bash
# Round-5 trip 3: the new GPU tests, the D = 16 scaling rehearsal, the default bench (MERA chi = 64 through the general
# machinery, the D = 16 sliced network).  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_svd_band.py -m gpu -q --timeout 600 \
  -k "sliced or mera or policy or falls_back or same_on_every_rank" > gpurun_out/r5_pytest_new.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r5_pytest_new.log
timeout 600 python tools/rr_scaling_rehearsal.py 16 2 > gpurun_out/r5_rr_rehearsal_D16.jsonl 2> gpurun_out/r5_rr_rehearsal_D16.err; echo "rehearsal rc=$?"
tail -3 gpurun_out/r5_rr_rehearsal_D16.err
timeout 1200 python bench.py --steps 5 --warmup 2 > gpurun_out/r5_bench3.json 2> gpurun_out/r5_bench3.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r5_bench3.json; tail -5 gpurun_out/r5_bench3.err
