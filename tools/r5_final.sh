#!/bin/bash
# Round-5 evidence on the final tree, in this order: rocprofv3 stats + PMC passes of the headline command -> traffic
# json (copied into profiles/ on the box so that the bench line of THIS run cites it) -> the whole GPU suite -> the
# driver's bench command -> (when the reference was shipped beside the repo: tools/reference_dropin/
# gpurun_with_reference.sh) the reference's own test files on backend="hip" -> a kernel table of the D = 16 sliced network.
# usage: tools/reference_dropin/gpurun_with_reference.sh --timeout 3000 -- 'bash tools/r5_final.sh'
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
O=$OUT/r5final; mkdir -p $O
rm -rf $OUT/prof_stats $OUT/prof_pmc_* $OUT/prof_rr16
if [ -d "$PWD/_reference_scratch/tensornetwork" ]; then export TN_REFERENCE_DIR=$PWD/_reference_scratch; fi
BENCH_ARGS="--steps 5" bash tools/profile.sh > $O/profile.log 2>&1; tail -6 $O/profile.log
python tools/prof_summary.py $OUT $O/prof_summary.txt > /dev/null 2>&1; head -14 $O/prof_summary.txt
python tools/traffic_json.py $OUT $OUT/bench_detail.json $O/traffic.json > /dev/null && cp $O/traffic.json profiles/r05_traffic.json
cat $O/traffic.json | head -20
rm -f $OUT/prof_stats/*kernel_trace.csv $OUT/prof_stats/*/*kernel_trace.csv
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
cp $OUT/bench_detail.json $O/bench_detail.json; tail -c 4000 $O/bench.out; echo
if [ -n "${TN_REFERENCE_DIR:-}" ]; then
  OUT=$O/refdropin PER_FILE_TIMEOUT=600 bash tools/reference_dropin/run_reference_tests.sh > $O/refdropin.log 2>&1; echo "refdropin rc=$?"; tail -14 $O/refdropin.log
  OUT=$PWD/gpurun_out
fi
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_rr16 -o rr -- python $OUT/../tools/rr64_probe.py --D 16 --max-slices 256 > $O/rr16_prof.log 2>&1; echo "rr16 prof rc=$?")
python tools/kernel_stats.py $OUT/prof_rr16 > $O/rr16_kernel_stats.txt 2>/dev/null; head -20 $O/rr16_kernel_stats.txt
find $OUT/prof_rr16 -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r5final/bench_detail.json").read())
print("value", r["value"], "kernel_ms", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], r["roofline"]["traffic_source"])
print({k: v.get("ok") for k, v in r["verified"].items() if isinstance(v, dict)})
print("cpu", r["cpu_baseline"]["kind"], r["cpu_baseline"]["value"])
m = r["mera_chi64"]; print("mera64", {k: m.get(k) for k in ("measured_slices", "measured_seconds", "measured_tflops", "measured_executed_macs")}, m.get("verified_runs"))
for pl, run in (m.get("measured") or {}).items():
  print(pl, run["slices_done"], run["seconds"], run["tflops"], run["energy_partial_sum"], run["stage_runs"], run.get("classes_kept_for_all_values"))
s = r["sliced_network"]; print("sliced", s["seconds"], s["tflops"], s["mode"], s.get("executed_equals_model"))
PY
