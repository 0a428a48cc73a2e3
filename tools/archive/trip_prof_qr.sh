#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cat > /tmp/qr1.py <<PY
import sys, time, numpy as np
sys.path.insert(0, "$R")
import tensornetwork_amd as ta
be = ta.get_hip_backend()
m, n = int(sys.argv[1]), int(sys.argv[2])
x = be.device_random((m, n), dtype=np.float32, seed=1)
be.qr(x, 1); be.synchronize()
t0 = time.perf_counter(); be.qr(x, 1); be.synchronize(); print("qr", m, n, time.perf_counter() - t0)
PY
cd /tmp
for shape in "4096 4096" "65536 256"; do
  tag=$(echo $shape | tr ' ' 'x')
  rm -rf $OUT/prof_stats_qr$tag
  rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_qr$tag -o qr -- python /tmp/qr1.py $shape > $OUT/qr$tag.log 2>&1
  grep "^qr" $OUT/qr$tag.log
done
cd $R
python - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob('gpurun_out/prof_stats_qr*/*.db')):
  print(db)
  c = sqlite3.connect(db)
  for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:8]:
    print(f"{calls:6d} {total/1e3:10.3f} ms {avg/1e3:9.3f} ms {pct:6.2f}%  {name[:110]}")
PY
