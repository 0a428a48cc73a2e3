#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -k "gemm or tensordot" > gpurun_out/pytest_gemm.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gemm.log
timeout 600 python tools/gemm_sweep.py --skinny > gpurun_out/skinny.jsonl 2> gpurun_out/skinny.err; echo "skinny rc=$?"
cat gpurun_out/skinny.jsonl | python -c "
import sys, json
for l in sys.stdin:
  r = json.loads(l); print('%-28s %-20s %8d %8d %8d  %8.3f ms %8.1f TF %8.1f GB/s' % (r['kernel'], r['variant'], r['m'], r['n'], r['k'], r['ms'], r['tflops'], r['gbps']))
"
tail -3 gpurun_out/skinny.err
timeout 600 python tools/rr64_probe.py --D 12 --min-slices 64 --max-slices 8 2>&1 | tail -1
