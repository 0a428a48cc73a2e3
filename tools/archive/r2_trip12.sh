#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== gemm tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_linalg.py -m gpu -q --timeout 600 -k "gemm or f64 or complex or qr or eigh or expm" > $OUT/pytest_k.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest_k.log
echo "== f32/f64 probe"
timeout 300 python tools/f32_gemm_probe.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
  try: r = json.loads(l)
  except ValueError: print(l.strip()); continue
  if r['dtype'] == 'f64' or r['m'] == 4096: print(r['dtype'], r['tA'], r['tB'], r['m'], r['n'], r['k'], round(r['tflops'], 1), r['kernel'])"
echo "== power f64"
timeout 200 python tools/power_probe.py --dtype f64 --seconds 2 --shapes 4096x4096x4096,8192x8192x2048 --fills zeros,normal > $OUT/power_f64.jsonl 2>&1; cut -c1-600 $OUT/power_f64.jsonl
timeout 200 python tools/power_probe.py --dtype f32 --seconds 2 --shapes 4096x4096x4096 --fills zeros,normal > $OUT/power_f32.jsonl 2>&1; cut -c1-400 $OUT/power_f32.jsonl
