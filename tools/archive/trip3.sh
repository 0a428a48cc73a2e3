#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
rm -rf $OUT/prof_stats $OUT/prof_pmc_*
bash tools/profile.sh > $OUT/profile.log 2>&1; tail -8 $OUT/profile.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_svd -o svd -- python $OLDPWD/tools/svd_probe.py --check 0 --sizes 4096 --reps 0 > $OUT/prof_svd.log 2>&1
cd $OLDPWD
TNH_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 1 --warmup 1 --svd-n 0 --no-cpu-baseline --bond 64 --rr-bond 8 2>&1 | grep -c metric
