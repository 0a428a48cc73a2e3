#!/bin/bash
# Round 4, trip 8: the reference's own test files on backend="hip" (final tree), the SVD / linalg GPU tests after the
# last f64 tweaks, f32 / f64 band timings, two-site DMRG sweeps in f32 / f64.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t8; mkdir -p $O
OUT=$O/refdropin TN_REFERENCE_DIR=$PWD/_reference_scratch bash tools/reference_dropin/run_reference_tests.sh > $O/refdropin.log 2>&1; tail -14 $O/refdropin.log
timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py tests/test_gpu_reference_dropin.py -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python tools/svd_sizes_probe.py > $O/svd_sizes.jsonl 2> $O/svd_sizes.err; cat $O/svd_sizes.jsonl | cut -c1-200
for dt in float32 float64; do
  timeout 400 python tests/perf_dmrg.py --bonds 256,512 --dtype $dt --cpu-max 0 >> $O/dmrg.txt 2>&1
  TNH_SVD_BAND64=0 true
done
tail -8 $O/dmrg.txt
