#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
python tools/d96_trace.py L0
TNH_GEMM_TAIL_SPLIT=0 python tools/d96_trace.py L0
cd /tmp; rm -rf $OUT/prof8
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof8 -o p -- python $R/tools/d96_trace.py L0 > $OUT/prof8.log 2>&1; echo "rc=$?"
cd $R
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob('gpurun_out/prof8/*.db')[0])
for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:8]:
  print(f"{calls:7d} {total/1e3:10.2f} ms {avg/1e3:9.4f} ms {pct:6.2f}  {name[:100]}")
PY
