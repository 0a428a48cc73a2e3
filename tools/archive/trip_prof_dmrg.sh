#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python -m cProfile -o gpurun_out/dmrg.prof tests/perf_dmrg.py --n 24 --bonds ${DMRG_D:-256} --cpu-max 0 | tail -2
python -c "
import pstats
p = pstats.Stats('gpurun_out/dmrg.prof'); p.sort_stats('tottime').print_stats(14)
" 2>&1 | tail -24
cd /tmp; rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_stats_dmrg -o d -- python $OLDPWD/tests/perf_dmrg.py --n 24 --bonds ${DMRG_D:-256} --cpu-max 0 > /dev/null 2>&1; cd $OLDPWD
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/prof_stats_dmrg/*.db')[0]
c = sqlite3.connect(db)
for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:10]:
  print(f"{calls:7d} {total/1e3:10.1f} ms {avg/1e3:8.3f} ms {pct:6.2f}%  {name[:90]}")
PY
