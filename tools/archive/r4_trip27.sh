#!/bin/bash
# K-loop kernel with two boxes of prefetch at 48 rows: parity, then the D = 12 network.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t27; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gather_k_loop" --timeout 90 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 100 python tools/rr64_probe.py --D 12 --max-slices 8 > $O/rr64.json 2> $O/rr64.err; echo "rr64 rc=$?"; python -c "import json;r=json.load(open('$O/rr64.json'));print(r['sec_per_slice'],r['tflops'])"; tail -2 $O/rr64.err
