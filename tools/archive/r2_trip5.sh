#!/bin/bash
# Round-2 trip 5: SVD -- de Rijk sorting, early stop, grouped multi-stream rounds; parity of the SVD tests.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== svd tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_linalg.py tests/test_gpu_mps.py -m gpu -q --timeout 600 -k "svd or eigh or inv or dmrg or split or mps" > $OUT/pytest_svd.log 2>&1; echo "rc=$?"; tail -8 $OUT/pytest_svd.log
echo "== svd probe 4096 gauss"
timeout 600 python tools/svd_sweep_probe.py --n 4096 > $OUT/svd_sweep_4096.jsonl 2> $OUT/svd_sweep_4096.err; echo "rc=$?"; cat $OUT/svd_sweep_4096.jsonl; grep "tnh svd" $OUT/svd_sweep_4096.err | head -40
echo "== svd probe 2048 graded"
timeout 600 python tools/svd_sweep_probe.py --n 2048 --spectrum graded > $OUT/svd_sweep_2048g.jsonl 2> $OUT/svd_sweep_2048g.err; echo "rc=$?"; cat $OUT/svd_sweep_2048g.jsonl; grep "tnh svd" $OUT/svd_sweep_2048g.err | tail -45
echo "== svd probe tool (parity on odd shapes)"
timeout 600 python tools/svd_probe.py --sizes 1024,2048 > $OUT/svd_probe.log 2>&1; tail -15 $OUT/svd_probe.log
