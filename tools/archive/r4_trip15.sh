#!/bin/bash
# Gather GEMM (tnh_gemm_gather): parity tests, then the D = 12 north-star network with and without it.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t15; mkdir -p $O
timeout 150 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k gather --timeout 140 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
for g in 0 1; do
  TNH_GATHER_GEMM=$g timeout 100 python tools/rr64_probe.py --D 12 --max-slices 8 > $O/rr64_gather$g.json 2> $O/rr64_gather$g.err; echo "rr64 gather=$g rc=$?"; cat $O/rr64_gather$g.json; tail -3 $O/rr64_gather$g.err
done
cd /tmp && TNH_GATHER_GEMM=1 timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof -o rr -- python $GRAFT_REPO_ROOT/tools/rr64_probe.py --D 12 --max-slices 4 > $O/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT && python tools/kernel_stats.py $O/prof "rocprofv3 --kernel-trace --stats -- TNH_GATHER_GEMM=1 python tools/rr64_probe.py --D 12 --max-slices 4 (warm-up + timed pass: 8 slices; MI355X, round 4)" > $O/rr64_D12_gather_kernel_stats.txt 2>&1; head -16 $O/rr64_D12_gather_kernel_stats.txt
