#!/bin/bash
# rocprofv3 PMC passes (one run per counter group, no trace domains) of an arbitrary command; per-kernel averages
# of every counter are printed to <out>/<tag>_pmc.txt.   tools/pmc_passes.sh <tag> "<cmd>" "<ctr group 1>" "<ctr group 2>" ...
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=$1; CMD=$2; shift 2
mkdir -p $OUT
ROOT=$PWD
cd /tmp
i=0
for ctr in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr -d $OUT/pmc_${TAG}_$i -o run -- bash -c "cd $ROOT && $CMD" > $OUT/pmc_${TAG}_$i.log 2>&1
  echo "pmc [$ctr] rc=$?"
done
cd $ROOT
python - "$OUT" "$TAG" <<'PY'
import glob, os, sqlite3, sys
out, tag = sys.argv[1], sys.argv[2]
lines = []
for d in sorted(glob.glob(os.path.join(out, f"pmc_{tag}_*"))):
  if not os.path.isdir(d): continue
  for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
    c = sqlite3.connect(db)
    try:
      q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
           "group by kernel_name, counter_name order by counter_name, kernel_name")
      for name, ctr, n, avg in c.execute(q):
        if avg > 0 and "gemm" in name:
          lines.append(f"{ctr:36s} n={n:3d} avg={avg:18.1f}  {name[:110]}")
    except Exception as e:
      lines.append(f"# {db}: {e}")
open(os.path.join(out, f"{tag}_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
