#!/bin/bash
# D = 12 network: kernel tables of the classic lowering and of the gather lowering (pieces >= 256 B), same box.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t18; mkdir -p $O
prof() {  # tag, env...
  tag=$1; shift
  (cd /tmp && env "$@" timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o rr -- python $GRAFT_REPO_ROOT/tools/rr64_probe.py --D 12 --max-slices 4 > $O/prof_$tag.log 2>&1; echo "prof $tag rc=$?")
  python tools/kernel_stats.py $O/prof_$tag "rocprofv3 --kernel-trace --stats -- $* python tools/rr64_probe.py --D 12 --max-slices 4 (warm-up + timed pass: 8 slices; MI355X, round 4)" > $O/rr64_D12_${tag}_kernel_stats.txt 2>&1
  head -14 $O/rr64_D12_${tag}_kernel_stats.txt; tail -1 $O/rr64_D12_${tag}_kernel_stats.txt
}
prof classic TNH_GATHER_GEMM=0
prof gather_p256 TNH_GATHER_GEMM=1 TNH_GATHER_MIN_PIECE=256
