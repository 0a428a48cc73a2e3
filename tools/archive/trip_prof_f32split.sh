#!/bin/bash
# rocprofv3 kernel stats of the f32-on-bf16-cores path and of the 4-wave A/B kernel -> profiles/ summaries
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp
run() { tag=$1; shift; rm -rf $OUT/prof3_$tag; timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof3_$tag -o p -- "$@" > $OUT/prof3_$tag.log 2>&1; echo "$tag rc=$?"; }
run f32split python $R/tools/f32_split_probe.py --perf
run w4 python $R/tools/w4_probe.py --perf
cd $R
python - <<'PY'
import sqlite3, glob, os
out = []
for d in sorted(glob.glob('gpurun_out/prof3_*/')):
  dbs = glob.glob(d + '*.db')
  if not dbs: continue
  c = sqlite3.connect(dbs[0])
  out.append(f"# rocprofv3 --kernel-trace --stats   ({os.path.basename(d.rstrip('/'))})")
  out.append(f"{'calls':>7} {'total_ms':>10} {'avg_ms':>9} {'pct':>6}  kernel")
  for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:10]:
    out.append(f"{calls:7d} {total/1e3:10.2f} {avg/1e3:9.4f} {pct:6.2f}  {name[:120]}")
  out.append("")
open('gpurun_out/f32split_w4_stats.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -4 $OUT/prof3_f32split.log
