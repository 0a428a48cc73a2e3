#!/bin/bash
# rocprofv3 kernel stats of the secondary workloads (SVD top-k, QR, DMRG, sliced network) -> one summary
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp
run() { tag=$1; shift; rm -rf $OUT/prof2_$tag; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof2_$tag -o p -- "$@" > $OUT/prof2_$tag.log 2>&1; echo "$tag rc=$?"; }
run svd4096 python $R/tools/svd_probe.py --check 0 --sizes 4096 --reps 1
run qr python $R/tools/qr_one.py 4096 4096
run dmrg python $R/tests/perf_dmrg.py --n 24 --bonds 256 --cpu-max 0
run rr12 python $R/tools/rr64_probe.py --D 12 --min-slices 64 --max-slices 16
cd $R
python - <<'PY'
import sqlite3, glob, os
out = []
for d in sorted(glob.glob('gpurun_out/prof2_*/')):
  dbs = glob.glob(d + '*.db')
  if not dbs: continue
  c = sqlite3.connect(dbs[0])
  out.append(f"# rocprofv3 --kernel-trace --stats   ({os.path.basename(d.rstrip('/'))})")
  out.append(f"{'calls':>7} {'total_ms':>10} {'avg_ms':>9} {'pct':>6}  kernel")
  for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:9]:
    out.append(f"{calls:7d} {total/1e3:10.2f} {avg/1e3:9.4f} {pct:6.2f}  {name[:110]}")
  out.append("")
open('gpurun_out/secondary_stats.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
