#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
echo "== raster A/B on the headline shape"
for v in "bf16_256pp:r0" "bf16_256pp:r1"; do timeout 300 python tools/gemm_one.py --variant $v --m 65536 --n 65536 --k 65536 --iters 3 --fill uniform 2>&1 | tail -1; done
echo "== mera"; timeout 600 python tools/mera_probe.py --chi 4,8,16 2>&1 | tail -4
