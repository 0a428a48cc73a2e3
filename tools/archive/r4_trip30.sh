#!/bin/bash
# D = 12 slice with and without the K <= 16 store-stream kernel, same box.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t30; mkdir -p $O
for s in 0 1; do
  TNH_GEMM_SMALLK=$s timeout 40 python tools/rr64_probe.py --D 12 --max-slices 8 > $O/rr64_smallk$s.json 2> $O/rr64_smallk$s.err; echo "rr64 smallk=$s rc=$?"; python -c "import json;r=json.load(open('$O/rr64_smallk$s.json'));print(r['sec_per_slice'],r['tflops'])"; tail -2 $O/rr64_smallk$s.err
done
