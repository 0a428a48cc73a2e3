#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp; rm -rf $OUT/prof9
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof9 -o p -- python $R/tools/svd_probe.py --check 0 --sizes 2048 --reps 1 --dtype f64 > $OUT/prof9.log 2>&1; echo "rc=$?"
cd $R
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob('gpurun_out/prof9/*.db')[0])
for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:6]:
  print(f"{calls:7d} {total/1e3:10.2f} ms {avg/1e3:9.4f} ms {pct:6.2f}  {name[:100]}")
PY
