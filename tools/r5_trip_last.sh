#!/bin/bash
# round-5 last trip, on the final tree: the whole GPU suite, then the driver's bench command
# (product changes since tools/r5_final.sh ran: distributed.py grid partitions -- GPU-tested in
# profiles/r05_pytest_gpu_sliced_after_grid.txt --, the allocator's amortised collector passes, K1 ragged bricks)
mkdir -p gpurun_out/r5last
O=gpurun_out/r5last
timeout 600 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
cp gpurun_out/bench_detail.json $O/bench_detail.json; tail -c 3600 $O/bench.out; echo
