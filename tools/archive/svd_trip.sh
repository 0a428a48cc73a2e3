#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
echo "== default (EIG=4, INNER=1, CROSS=1) with checks"
timeout 600 python tools/svd_probe.py --check 1 --sizes 1024,2048,4096 2>&1 | tail -14
for cfg in "4 0" "3 1"; do
  set -- $cfg
  echo "== EIG=$1 CROSS=$2"
  TNH_SVD_EIG=$1 TNH_SVD_CROSS=$2 timeout 600 python tools/svd_probe.py --check 0 --sizes 2048,4096 2>&1 | tail -3
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_svd -o svd -- python $OLDPWD/tools/svd_probe.py --check 0 --sizes 4096 --reps 0 > $OUT/prof_svd.log 2>&1
cd $OLDPWD
python - <<'PY'
import glob, sqlite3
for db in glob.glob("gpurun_out/prof_svd/**/*.db", recursive=True):
  c = sqlite3.connect(db)
  for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:4]:
    print(f"{calls:6d} {total/1e3:12.3f} {avg/1e3:12.4f} {pct:7.2f}  {name[:80]}")
PY
timeout 900 python -m pytest tests -m gpu -q -k svd --timeout 600 2>&1 | tail -3
