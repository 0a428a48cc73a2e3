#!/bin/bash
# Round-3 evidence on the final tree: GPU tests, the default bench line, rocprofv3 stats + PMC passes of the
# headline command, and a kernel table of the MERA chi = 32 layer.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
rm -rf $OUT/prof_stats $OUT/prof_pmc_* $OUT/prof_stats_* $OUT/prof_mera
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
fi
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
bash tools/profile.sh > $OUT/profile.log 2>&1; tail -6 $OUT/profile.log
python tools/prof_summary.py $OUT $OUT/prof_summary.txt > /dev/null
python tools/traffic_json.py $OUT $OUT/bench.json $OUT/traffic.json > /dev/null; head -30 $OUT/traffic.json
head -12 $OUT/prof_summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mera -o mera -- python $OUT/../bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sweep --no-extras --no-verify --mera-chi 32 --svd-n 0 --rr-bond 0 > $OUT/prof_mera.log 2>&1; echo "mera prof rc=$?")
python tools/kernel_list.py $OUT/prof_mera/mera_kernel_trace.csv 300 150 > $OUT/mera_kernel_list.txt 2>&1; tail -70 $OUT/mera_kernel_list.txt
rm -f $OUT/prof_mera/*kernel_trace.csv $OUT/prof_stats/*kernel_trace.csv
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print("value", r["value"], "kernel_ms", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"])
for row in r.get("bond_sweep", []):
  print(row)
print("svd", r.get("svd", {}).get("seconds"), "rr", r.get("sliced_network", {}).get("seconds"), r.get("cpu_baseline"))
print("verified all_ok", r.get("verified", {}).get("all_ok"))
for h in r.get("helpers", []):
  print("%-80s %.0f GB/s" % (h["op"][:80], h["gbps"]))
PY
