#!/bin/bash
# Final tree: smoke(), then bench.py with only the headline, the sliced network and the extras' gather leg skipped/kept short.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t24; mkdir -p $O
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 150 python bench.py --steps 20 --warmup 5 --no-sweep --no-extras --svd-n 0 --mera-chi 0 --no-cpu-baseline > $O/bench_short.out 2> $O/bench_short.err; echo "bench rc=$?"; tail -1 $O/bench_short.out | cut -c1-1500; tail -2 $O/bench_short.err
