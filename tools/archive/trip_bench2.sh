#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
timeout 600 python bench.py --svd-n 0 --rr-bond 0 --no-cpu-baseline --no-sweep --mera-chi 0 --steps 4 --warmup 2 2>&1 | python -c "
import sys, json
for l in sys.stdin:
  if l.startswith('{'):
    r = json.loads(l); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'])
  else: print(l.strip()[:200])
"
done
cd /tmp; rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_stats_b2 -o bench -- python $OLDPWD/bench.py --svd-n 0 --rr-bond 0 --no-cpu-baseline --no-sweep --mera-chi 0 --steps 3 --warmup 1 > $OLDPWD/gpurun_out/b2.log 2>&1; cd $OLDPWD
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/prof_stats_b2/*.db')[0]
c = sqlite3.connect(db)
for row in c.execute("select * from top_kernels"): print(row)
PY
