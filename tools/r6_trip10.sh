#!/bin/bash
# Round 6, trip 10: what the driver runs -- smoke, the whole -m gpu suite, bench.py --gpus 1 --steps 20 --warmup 5.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/t10_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/t10_smoke.log | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/t10_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/t10_pytest_gpu.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/t10_bench.json 2> $OUT/t10_bench.err; echo "bench rc=$?"
tail -1 $OUT/t10_bench.json | cut -c1-4000; tail -3 $OUT/t10_bench.err
cp bench_detail.json $OUT/t10_bench_detail.json 2>/dev/null
