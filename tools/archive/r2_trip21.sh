#!/bin/bash
# do the two streams of the two-group SVD schedule overlap?  kernel trace -> timeline of a few rounds
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp; rm -rf $OUT/prof6
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof6 -o p -- python $R/tools/svd_probe.py --check 0 --sizes 4096 --reps 1 > $OUT/prof6.log 2>&1; echo "rc=$?"
cd $R
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob('gpurun_out/prof6/*.db')[0])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = list(c.execute("select name, start, end, queue_id, stream_id from kernels order by start"))
# take a window in the middle
mid = len(rows) // 2
t0 = rows[mid][1]
for name, st, en, q, sid in rows[mid:mid + 24]:
  short = "gram" if "gram" in name else ("eig" if "eig" in name else ("upd" if "update" in name else name[:12]))
  print(f"{short:5s} q{q} s{sid}  start {(st - t0) / 1e3:8.1f} us  dur {(en - st) / 1e3:6.1f} us")
# total busy time vs span
span = rows[-1][2] - rows[0][1]
busy = sum(e - s for _, s, e, _, _ in rows)
print("span ms", span / 1e6, "sum of kernel durations ms", busy / 1e6)
PY
