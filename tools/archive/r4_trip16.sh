#!/bin/bash
# Gather GEMM with two tiles of register prefetch: parity, then the D = 12 network: classic / gather / gather with a minimum piece size.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t16; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k gemm_gather --timeout 90 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 100 python tools/rr64_probe.py --D 12 --max-slices 8 > $O/rr64_$tag.json 2> $O/rr64_$tag.err; echo "rr64 $tag rc=$?"; cut -c1-220 $O/rr64_$tag.json; tail -2 $O/rr64_$tag.err
}
run classic TNH_GATHER_GEMM=0
run gather TNH_GATHER_GEMM=1
run gather_p256 TNH_GATHER_GEMM=1 TNH_GATHER_MIN_PIECE=256
run gather_p1024 TNH_GATHER_GEMM=1 TNH_GATHER_MIN_PIECE=1024
cd /tmp && TNH_GATHER_GEMM=1 timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof -o rr -- python $GRAFT_REPO_ROOT/tools/rr64_probe.py --D 12 --max-slices 4 > $O/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT && python tools/kernel_stats.py $O/prof "rocprofv3 --kernel-trace --stats -- TNH_GATHER_GEMM=1 python tools/rr64_probe.py --D 12 --max-slices 4 (warm-up + timed pass: 8 slices; two tiles of prefetch; MI355X, round 4)" > $O/rr64_D12_gather_kernel_stats.txt 2>&1; head -12 $O/rr64_D12_gather_kernel_stats.txt
