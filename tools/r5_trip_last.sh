#!/bin/bash
# round-5 last trip: the GPU tests that reach tensornetwork_amd/distributed.py (the only product file changed since
# tools/r5_final.sh ran) and the driver's bench command on the final tree
mkdir -p gpurun_out/r5last
O=gpurun_out/r5last
timeout 900 python -m pytest $(grep -l "distributed\|contract_sliced" tests/test_gpu*.py) -m gpu -q --timeout 600 > $O/pytest_gpu_distributed.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu_distributed.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
cp gpurun_out/bench_detail.json $O/bench_detail.json; tail -c 3500 $O/bench.out; echo
