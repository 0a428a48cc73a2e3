#!/bin/bash
# Round 4, trip 6: the driver's bench command on the final tree (default: one chi = 64 placement in full) and the
# rocprofv3 kernel tables of the f32 / f64 band SVD.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
O=$OUT/r4t6; mkdir -p $O
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
cp $OUT/bench_detail.json $O/bench_detail.json; tail -c 4000 $O/bench.out; echo
for dt in f32 f64; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_$dt -o svd -- python $OUT/../tools/svd_stats_run.py $dt > $O/svd_$dt.log 2>&1; echo "svd $dt prof rc=$?")
  f=$(find $OUT/prof_svd_$dt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/svd_${dt}_kernel_stats.csv
  find $OUT/prof_svd_$dt -name "*kernel_trace.csv" -delete
done
head -30 $O/svd_f32_kernel_stats.csv; head -34 $O/svd_f64_kernel_stats.csv
