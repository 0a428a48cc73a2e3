#!/bin/bash
# Round 4, trip 7: strip re-synchronisation of the k-major view GEMM (TNH_GEMM_KSYNC = K-tiles between meetings).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t7; mkdir -p $O
for ks in 0 512 256 128 64; do
  TNH_GEMM_KSYNC=$ks timeout 300 python tools/kmajor_touch_probe.py --shapes 8192x8192x262144,8192x8192x65536 --iters 6 2>> $O/err.txt | sed "s/^/ksync=$ks /" | tee -a $O/ksync.txt
done
TNH_GEMM_KSYNC=256 timeout 300 python tools/kmajor_touch_probe.py --shapes 8192x8192x262144 --iters 6 --zeros 2>> $O/err.txt | sed "s/^/ksync=256 /" | tee -a $O/ksync.txt
TNH_GEMM_KSYNC=0 timeout 300 python tools/kmajor_touch_probe.py --shapes 8192x8192x262144 --iters 6 --zeros 2>> $O/err.txt | sed "s/^/ksync=0 /" | tee -a $O/ksync.txt
tail -3 $O/err.txt
