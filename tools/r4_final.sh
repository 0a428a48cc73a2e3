#!/bin/bash
# Round-4 evidence on the final tree, in this order: rocprofv3 stats + PMC passes of the headline command -> traffic
# json (copied into profiles/ on the box so that the bench line of THIS run cites it) -> the whole GPU suite -> the
# driver's bench command (both chi = 64 placements in full) -> kernel tables of the f32 / f64 band SVD.
# usage: /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r4_final.sh'
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
O=$OUT/r4final; mkdir -p $O
rm -rf $OUT/prof_stats $OUT/prof_pmc_* $OUT/prof_svd_*
BENCH_ARGS="--steps 5" bash tools/profile.sh > $O/profile.log 2>&1; tail -6 $O/profile.log
python tools/prof_summary.py $OUT $O/prof_summary.txt > /dev/null 2>&1; head -14 $O/prof_summary.txt
python tools/traffic_json.py $OUT $OUT/bench_detail.json $O/traffic.json > /dev/null && cp $O/traffic.json profiles/r04_traffic.json
cat $O/traffic.json | head -20
rm -f $OUT/prof_stats/*kernel_trace.csv $OUT/prof_stats/*/*kernel_trace.csv
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --mera64-full ${MERA64:-1} --mera64-budget 400 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
cp $OUT/bench_detail.json $O/bench_detail.json; tail -c 4000 $O/bench.out; echo
for dt in f32 f64; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_$dt -o svd -- python $OUT/../tools/svd_stats_run.py $dt > $O/svd_$dt.log 2>&1; echo "svd $dt prof rc=$?")
  find $OUT/prof_svd_$dt -name "*kernel_trace.csv" -delete
done
python tools/svd_stats_summary.py $OUT $O
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4final/bench_detail.json").read())
print("value", r["value"], "kernel_ms", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], r["roofline"]["traffic_source"])
print({k: v.get("ok") for k, v in r["verified"].items() if isinstance(v, dict)})
m = r["mera_chi64"]; print("mera64", {k: m.get(k) for k in ("measured_slices", "measured_seconds", "measured_tflops", "layer_seconds_1gpu_extrapolated")}, m.get("verified_runs"))
for pl, run in (m.get("measured") or {}).items():
  print(pl, run["slices_done"], run["seconds"], run["tflops"], run["energy_partial_sum"])
PY
