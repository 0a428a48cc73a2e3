#!/bin/bash
# Gather GEMM: per-shape timings against permute + streaming GEMM (both orientations), then the D = 12 network.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t17; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k gemm_gather --timeout 90 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 120 python tools/gather_probe.py > $O/gather_probe.jsonl 2> $O/gather_probe.err; echo "probe rc=$?"; cat $O/gather_probe.jsonl; tail -3 $O/gather_probe.err
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 100 python tools/rr64_probe.py --D 12 --max-slices 8 > $O/rr64_$tag.json 2> $O/rr64_$tag.err; echo "rr64 $tag rc=$?"; cut -c1-220 $O/rr64_$tag.json; tail -2 $O/rr64_$tag.err
}
run classic TNH_GATHER_GEMM=0
run gather_p256 TNH_GATHER_GEMM=1 TNH_GATHER_MIN_PIECE=256
