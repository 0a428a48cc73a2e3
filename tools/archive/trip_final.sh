#!/bin/bash
# Round evidence: full bench line, rocprofv3 stats + PMC passes of the same command, summaries.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
rm -rf $OUT/prof_stats $OUT/prof_pmc_* $OUT/prof_stats_*
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
bash tools/profile.sh > $OUT/profile.log 2>&1; tail -6 $OUT/profile.log
python tools/prof_summary.py $OUT $OUT/prof_summary.txt > /dev/null
python tools/traffic_json.py $OUT $OUT/bench.json $OUT/traffic.json > /dev/null; cat $OUT/traffic.json | head -30
head -12 $OUT/prof_summary.txt
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print("value", r["value"], "kernel_ms", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"])
for row in r.get("bond_sweep", []):
  print(row)
print("svd", r.get("svd", {}).get("seconds"), "rr", r.get("sliced_network", {}).get("seconds"), r.get("cpu_baseline"))
PY
