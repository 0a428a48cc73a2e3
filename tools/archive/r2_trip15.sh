#!/bin/bash
# Round-2 trip 15: per-dispatch picture of one slice of the D=12 network (which launches carry the time now).
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
rm -rf $OUT/prof3_rr12
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof3_rr12 -o p -- python $R/tools/rr64_probe.py --D 12 --min-slices 64 --max-slices 16 > $OUT/prof3_rr12.log 2>&1; echo "rc=$?"
cd $R
python - <<'PY'
import sqlite3, glob, collections
db = glob.glob('gpurun_out/prof3_rr12/*.db')[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kernel' in t.lower()][:20])
for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:14]:
  print(f"{calls:7d} {total/1e3:10.2f} {avg/1e3:9.4f} {pct:6.2f}  {name[:110]}")
# per-dispatch: group by (kernel, grid, workgroup) 
try:
  q = "select name, grid_x, grid_y, workgroup_x, count(*), sum(duration), avg(duration) from kernels group by name, grid_x, grid_y order by sum(duration) desc limit 40"
  rows = list(c.execute(q))
except Exception as e:
  print("kernels view failed:", e)
  cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
  print(cols)
  rows = []
for name, gx, gy, wx, n, tot, avg in rows:
  print(f"{n:6d} {tot/1e6:9.2f} ms {avg/1e3:9.1f} us  grid {gx}x{gy} wg {wx}  {name[:90]}")
PY
