#!/bin/bash
# Round 4, trip 11: bench.py's distributed path through the K8 RCCL communicator in a 1-rank group (the bring-up
# watchdog, the helper-thread ncclCommInitRank, barriers, max-over-ranks, the sliced network's all-reduce).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t11; mkdir -p $O
TNH_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-sweep --no-extras --svd-n 0 --mera-chi 0 --no-cpu-baseline > $O/bench_dist1.out 2> $O/bench_dist1.err; echo "rc=$?"; tail -5 $O/bench_dist1.err; cat $O/bench_dist1.out | cut -c1-1500
