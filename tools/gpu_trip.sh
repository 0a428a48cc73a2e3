#!/bin/bash
# One GPU-box visit: tests, smoke, bench, kernel sweep, rocprof. Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" | tee gpurun_out/trip.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/trip.log
echo "== pytest" | tee -a gpurun_out/trip.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/trip.log
tail -30 gpurun_out/pytest_gpu.log
echo "== sweep" | tee -a gpurun_out/trip.log
timeout 600 python tools/gemm_sweep.py ${SWEEP_ARGS:-} > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err; echo "sweep rc=$?" | tee -a gpurun_out/trip.log
cat gpurun_out/sweep.jsonl | head -60
echo "== bench" | tee -a gpurun_out/trip.log
timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" | tee -a gpurun_out/trip.log
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
