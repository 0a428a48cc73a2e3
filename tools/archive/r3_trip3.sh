#!/bin/bash
# round 3 band-SVD trips: tests, probe, kernel table.  usage: bash tools/r3_trip3.sh <tag> [quick]
set -u
O=gpurun_out/${1:-r3tx}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_svd_band.py -q --timeout 600 > $O/pytest_band.log 2>&1; echo "pytest rc=$?" | tee $O/trip.log
tail -25 $O/pytest_band.log
for args in "1024 64 gauss" "2048 128 gauss" "4096 256 gauss" "4096 256 graded"; do
  timeout 300 python tools/svd_band_probe.py $args >> $O/probe.jsonl 2>> $O/probe.err; echo "probe $args rc=$?" | tee -a $O/trip.log
done
python - <<PY
import json
for l in open("$O/probe.jsonl"):
  r=json.loads(l); print(r["n"],r["kind"],"factor %.1f vectors %.1f total %.1f backend %.1f"%(r["rep2"]["factor_ms"],r["rep2"]["vectors_ms"],r["rep2"]["total_ms"],r["backend_svd_ms"]),r["rep2"]["status"],r["path"],"s_err %.2e orth %.2e %.2e resid %.2e"%(r.get("s_err_over_s0",0),r.get("orth_u",0),r.get("orth_v",0),r.get("resid_over_s0",0)))
PY
tail -5 $O/probe.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o band4096 -- python $GRAFT_REPO_ROOT/tools/svd_band_probe.py 4096 256 gauss --no-check > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/$O/trip.log
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:16]:
  print("%-60s n=%6s total/call=%8.2f ms avg=%9.1f us min=%8.1f max=%8.1f"%(r["Name"][:60],r["Calls"],float(r["TotalDurationNs"])/5e6,float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
rm -f $O/prof/*kernel_trace.csv
