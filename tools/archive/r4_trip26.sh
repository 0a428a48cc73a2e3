#!/bin/bash
# Final tree: bench.py's sliced-network leg alone (all 144 slices, every partial against f32).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t26; mkdir -p $O
timeout 100 python tools/rr64_check.py > $O/rr64_check.json 2> $O/rr64_check.err; echo "check rc=$?"; cat $O/rr64_check.json; tail -3 $O/rr64_check.err
