#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== f32 v2 tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "f32 or gemm_all_layouts or tail" > $OUT/pytest_k.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_k.log
echo "== f32 probe"
timeout 300 python tools/f32_gemm_probe.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
  try: r = json.loads(l)
  except ValueError: print(l.strip()); continue
  print(r['dtype'], r['tA'], r['tB'], r['m'], r['n'], r['k'], round(r['tflops'], 1), r['kernel'])"
echo "== power f32 / f64"
timeout 200 python tools/power_probe.py --dtype f32 --seconds 2 --shapes 4096x4096x4096 --fills zeros,normal 2>&1 | cut -c1-700
timeout 200 python tools/power_probe.py --dtype f64 --seconds 2 --shapes 4096x4096x4096 --fills zeros,normal 2>&1 | cut -c1-700
