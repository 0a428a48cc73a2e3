#!/bin/bash
# band SVD: tests, trace, and the bench's config-3 sweep
set -u
O=gpurun_out/${1:-r3t9}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_svd_band.py -q -x --timeout 600 > $O/pytest_band.log 2>&1; echo "pytest rc=$?" | tee $O/trip.log
tail -5 $O/pytest_band.log
timeout 300 python tools/svd_band_probe.py 4096 256 gauss > $O/probe_g4096.json 2>> $O/probe.err
python -c "
import json; r=json.load(open('$O/probe_g4096.json')); print('factor %.1f vectors %.1f total %.1f'%(r['rep2']['factor_ms'],r['rep2']['vectors_ms'],r['rep2']['total_ms']), r['rep2']['status'], 's_err %.2e orth %.2e'%(r['s_err_over_s0'], r['orth_u']))"
TNH_SVDB_BTFORK=0 timeout 300 python tools/svd_band_probe.py 4096 256 gauss --no-check > $O/probe_nofork.json 2>> $O/probe.err
python -c "
import json; r=json.load(open('$O/probe_nofork.json')); print('nofork: factor %.1f vectors %.1f total %.1f'%(r['rep2']['factor_ms'],r['rep2']['vectors_ms'],r['rep2']['total_ms']))"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o band4096 -- python $GRAFT_REPO_ROOT/tools/svd_band_probe.py 4096 256 gauss --no-check > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/$O/trip.log
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $O/prof/band4096_kernel_trace.csv | grep -v "mean=   0.0\|mean=  -0.0" | tee $O/trace_summary.txt
f=$O/prof/band4096_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:16]:
  print("%-60s n=%6s total/call=%8.2f ms avg=%9.1f us min=%8.1f max=%8.1f"%(r["Name"][:60],r["Calls"],float(r["TotalDurationNs"])/5e6,float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
rm -f $O/prof/*kernel_trace.csv
timeout 120 python tools/mixed_order_probe.py 2>&1 | tail -4
timeout 900 python bench.py --steps 2 --warmup 1 --bond 64 --no-sweep --no-extras --mera-chi 0 --rr-bond 0 --no-cpu-baseline > $O/bench_svd.json 2> $O/bench_svd.err; echo "bench rc=$?" | tee -a $O/trip.log
python - <<PY
import json
r=json.loads(open("$O/bench_svd.json").read().strip().splitlines()[-1])
s=r.get("svd",{})
print({k:v for k,v in s.items() if k!="sweep"})
for row in s.get("sweep",[]):
  print(row["n"],row["input"],row["order"],"%.4f s"%row["seconds"],"%.2f GB/s"%row["gbps"],row["path"],row.get("check"))
print(r.get("verified",{}).get("svd"))
PY
tail -3 $O/bench_svd.err
