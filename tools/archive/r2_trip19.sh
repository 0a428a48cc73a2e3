#!/bin/bash
# eig kernel time vs number of inner rounds (timing probe: TNH_SVD_SORT=16*k halves the rounds k times)
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
for k in 0 1 2 5; do
  cd /tmp; rm -rf $OUT/prof5_$k
  TNH_SVD_SORT=$((16*k)) TNH_SVD_MAXSWEEPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof5_$k -o p -- python $R/tools/svd_probe.py --check 0 --sizes 4096 --reps 1 > $OUT/prof5_$k.log 2>&1
  cd $R
  python - $k <<'PY'
import sqlite3, glob, sys
c = sqlite3.connect(glob.glob(f'gpurun_out/prof5_{sys.argv[1]}/*.db')[0])
for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:3]:
  print("halvings", sys.argv[1], f"{calls:7d} {total/1e3:10.2f} {avg/1e3:9.4f}  {name[:60]}")
PY
done
