#!/bin/bash
# K <= 16 store-stream kernel: values against float64 and timing against the tile kernel, then the D = 12 slice with it.
set -u
export TMPDIR=/tmp TNH_GEMM_SMALLK=1
O=$PWD/gpurun_out/r4t29; mkdir -p $O
timeout 60 python tools/smallk_check.py > $O/smallk_check.jsonl 2> $O/smallk_check.err; echo "check rc=$?"; cat $O/smallk_check.jsonl; tail -3 $O/smallk_check.err
timeout 40 python tools/rr64_probe.py --D 12 --max-slices 4 > $O/rr64_smallk.json 2> $O/rr64_smallk.err; echo "rr64 rc=$?"; python -c "import json;r=json.load(open('$O/rr64_smallk.json'));print(r['sec_per_slice'],r['tflops'])"; tail -2 $O/rr64_smallk.err
