#!/bin/bash
# Round 4, trip 12: strong-scaling rehearsal of the north-star network on ONE GPU (rank 0's share for world = 1 / 2 / 4 / 8).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t12; mkdir -p $O
timeout 300 python tools/rr_scaling_probe.py 12 > $O/rr_scaling.jsonl 2> $O/err.txt; cat $O/rr_scaling.jsonl; tail -2 $O/err.txt
