#!/bin/bash
# Round-2 trip 18: LDS eigensolver with broadcast rotations: SVD tests, A/B (bcast x workgroup size), kernel table.
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_linalg.py -m gpu -q --timeout 600 > $OUT/pytest_svd.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_svd.log
echo "== A/B"
for b in 1 0; do for nt in 1024 512 256; do
  echo "bcast=$b eig_nt=$nt: $(TNH_SVD_BCAST=$b TNH_SVD_EIGNT=$nt python tools/svd_probe.py --check 1 --sizes 4096,512 --reps 2 2>&1 | tail -2 | tr '\n' ' ')"
done; done
echo "== kernel table (bcast 1, 1024)"
cd /tmp; rm -rf $OUT/prof4_svd; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof4_svd -o p -- python $R/tools/svd_probe.py --check 0 --sizes 4096 --reps 1 > $OUT/prof4_svd.log 2>&1; cd $R
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob('gpurun_out/prof4_svd/*.db')[0])
for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:4]:
  print(f"{calls:7d} {total/1e3:10.2f} {avg/1e3:9.4f} {pct:6.2f}  {name[:100]}")
PY
