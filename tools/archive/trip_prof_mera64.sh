#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; rm -rf $OUT/prof2_mera64
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof2_mera64 -o p -- python $R/tools/mera_slice_sample.py --chi 64 --reps 0 > $OUT/prof2_mera64.log 2>&1; echo rc=$?
cd $R
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/prof2_mera64/*.db')[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, grid_x, (end-start)/1000.0 as us from kernels order by start"))
# split by placement: two contractions; print the big kernels in order
for name, gx, us in rows:
  if us > 2000: print(f"{us/1000:9.2f} ms  grid={gx:<10d} {name[:110]}")
PY
