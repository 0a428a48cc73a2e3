#!/bin/bash
# round-5 late trip: the staged-contraction tests on the GPU after the deterministic cache release / grid partition,
# then the 8-rank rehearsal of the chi = 64 MERA placement (all ranks' shares on this one GPU)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_workloads.py -m gpu -x -q -k "sliced or mera" > gpurun_out/r5_pytest_grid.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r5_pytest_grid.log
timeout 500 python tools/mera_rank_share_rehearsal.py 64 8 left 1 > gpurun_out/r5_mera_rehearsal2.jsonl 2> gpurun_out/r5_mera_rehearsal2.err
echo "rehearsal rc=$?"; tail -c 1200 gpurun_out/r5_mera_rehearsal2.jsonl; tail -3 gpurun_out/r5_mera_rehearsal2.err
