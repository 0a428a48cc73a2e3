#!/bin/bash
# Round-6 evidence on the final tree, in this order: rocprofv3 stats + PMC passes of the headline command -> traffic json
# (copied into profiles/ on the box so that the bench line of THIS run cites it) -> the whole GPU suite -> the driver's
# bench command -> kernel tables of the band SVD (f32 and f64, 4096^2 keep 256) -> the MPS chain's launches.
# usage: gpurun --timeout 4800 -- 'bash tools/r6_final.sh'
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
O=$OUT/r6final; mkdir -p $O
rm -rf $OUT/prof_stats $OUT/prof_pmc_* $OUT/prof_svd_f32 $OUT/prof_svd_f64
BENCH_ARGS="--steps 5" bash tools/profile.sh > $O/profile.log 2>&1; tail -6 $O/profile.log
python tools/prof_summary.py $OUT $O/prof_summary.txt > /dev/null 2>&1; head -14 $O/prof_summary.txt
python tools/traffic_json.py $OUT $OUT/bench_detail.json $O/traffic.json > /dev/null && cp $O/traffic.json profiles/r06_traffic.json
cat $O/traffic.json | head -20
rm -f $OUT/prof_stats/*kernel_trace.csv $OUT/prof_stats/*/*kernel_trace.csv
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
cp $OUT/bench_detail.json $O/bench_detail.json; tail -c 4200 $O/bench.out; echo
for dt in f32 f64; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_$dt -o svd -- python $OUT/../tools/svd_stats_run.py $dt > $O/svd_$dt.log 2>&1; echo "svd $dt prof rc=$?")
  find $OUT/prof_svd_$dt -name "*kernel_trace.csv" -delete
done
python tools/svd_stats_summary.py $OUT $O > /dev/null; head -24 $O/svd_band_f32_kernel_stats.txt
timeout 300 python tools/mps_chain_shapes.py > $O/mps_chain_shapes.jsonl 2>&1; tail -1 $O/mps_chain_shapes.jsonl
timeout 300 python tools/svd_fast_probe.py > $O/svd_fast_probe.jsonl 2>&1; grep -c '"ok": true' $O/svd_fast_probe.jsonl
