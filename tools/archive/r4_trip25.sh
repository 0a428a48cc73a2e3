#!/bin/bash
# Tail split over up to 32 tile rows: the tail-split tests, then the D = 12 network with and without it.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t25; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tail_split" --timeout 90 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for t in 0 1; do
  TNH_GEMM_TAIL_SPLIT=$t timeout 100 python tools/rr64_probe.py --D 12 --max-slices 8 > $O/rr64_tail$t.json 2> $O/rr64_tail$t.err; echo "rr64 tail=$t rc=$?"; python -c "import json;r=json.load(open('$O/rr64_tail$t.json'));print(r['sec_per_slice'],r['tflops'])"; tail -2 $O/rr64_tail$t.err
done
