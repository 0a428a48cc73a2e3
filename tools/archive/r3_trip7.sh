#!/bin/bash
# per-launch trace of the band SVD at 4096 (durations, gaps, per-panel samples)
set -u
O=gpurun_out/${1:-r3t7}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_svd_band.py -q -x --timeout 600 > $O/pytest_band.log 2>&1; echo "pytest rc=$?" | tee $O/trip.log
tail -5 $O/pytest_band.log
timeout 300 python tools/svd_band_probe.py 4096 256 gauss > $O/probe_g4096.json 2>> $O/probe.err
python -c "
import json; r=json.load(open('$O/probe_g4096.json')); print('factor %.1f vectors %.1f total %.1f'%(r['rep2']['factor_ms'],r['rep2']['vectors_ms'],r['rep2']['total_ms']), r['rep2']['status'], 's_err %.2e orth %.2e'%(r['s_err_over_s0'], r['orth_u']))"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o band4096 -- python $GRAFT_REPO_ROOT/tools/svd_band_probe.py 4096 256 gauss --no-check > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/$O/trip.log
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $O/prof/band4096_kernel_trace.csv | tee $O/trace_summary.txt
f=$O/prof/band4096_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:18]:
  print("%-60s n=%6s total/call=%8.2f ms avg=%9.1f us min=%8.1f max=%8.1f"%(r["Name"][:60],r["Calls"],float(r["TotalDurationNs"])/5e6,float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
rm -f $O/prof/*kernel_trace.csv
