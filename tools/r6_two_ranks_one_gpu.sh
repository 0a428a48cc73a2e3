#!/bin/bash
# Round 6: bench.py's N = 2 code path rehearsed on the ONE GPU of the box (both ranks on device 0).  RCCL refuses two
# ranks on one device, so (1) without --allow-host-exchange the run must END with exit code 3, (2) with it the headline,
# the sliced network and the strong_scaling / rccl_ranks fields go through the labelled host exchange.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TNH_BENCH_FORCE_DIST=1 TNHIP_DEVICE=0 TNH_COMM_INIT_TIMEOUT_S=60
ARGS="--gpus 2 --steps 2 --warmup 1 --bond 128 --no-sweep --no-extras --svd-n 0 --mera-chi 0 --no-cpu-baseline --rr-bond 12"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py $ARGS > $OUT/two_ranks_strict.out 2> $OUT/two_ranks_strict.err; echo "strict rc=$?"
grep -h "did not come up" $OUT/two_ranks_strict.err | head -2 | cut -c1-300
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py $ARGS --allow-host-exchange > $OUT/two_ranks_host.out 2> $OUT/two_ranks_host.err; echo "host-exchange rc=$?"
tail -1 $OUT/two_ranks_host.out | cut -c1-3000; tail -4 $OUT/two_ranks_host.err | cut -c1-300
