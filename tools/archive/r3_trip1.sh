#!/bin/bash
# round 3, trip 1: first run of the band SVD (K7b): correctness tests, phase timings, kernel table
set -u
mkdir -p gpurun_out/r3t1
export TMPDIR=/tmp
O=gpurun_out/r3t1
timeout 900 python -m pytest tests/test_gpu_svd_band.py -x -q --timeout 600 > $O/pytest_band.log 2>&1; echo "pytest rc=$?" | tee $O/trip.log
tail -40 $O/pytest_band.log
for args in "1024 64 gauss" "2048 128 gauss" "4096 256 gauss" "4096 256 graded"; do
  timeout 300 python tools/svd_band_probe.py $args >> $O/probe.jsonl 2>> $O/probe.err; echo "probe $args rc=$?" | tee -a $O/trip.log
done
cat $O/probe.jsonl; tail -5 $O/probe.err
TNH_SVDB_DPP=0 timeout 300 python tools/svd_band_probe.py 4096 256 gauss --no-check >> $O/probe_nodpp.jsonl 2>> $O/probe.err
cat $O/probe_nodpp.jsonl
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o band4096 -- python $GRAFT_REPO_ROOT/tools/svd_band_probe.py 4096 256 gauss --no-check > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/$O/trip.log
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head -3
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f"
