#!/bin/bash
# D = 12 slice permutes: rates, and HBM-side bytes (FETCH_SIZE / WRITE_SIZE) against the algorithmic bytes
set -u
O=$PWD/gpurun_out/${1:-r3t33}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/permute_set_probe.py > $O/rates.jsonl 2>> $O/err.txt
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$ctr -o p -- python $GRAFT_REPO_ROOT/tools/permute_set_probe.py > $O/pmc_$ctr.log 2>&1
  echo "pmc $ctr rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, json, glob
rates=[json.loads(l) for l in open("$O/rates.jsonl")]
def series(ctr):
  f=glob.glob("$O/pmc_%s/*counter_collection.csv" % ctr)[0]
  rows=[r for r in csv.DictReader(open(f)) if "permute" in r["Kernel_Name"] or "gather" in r["Kernel_Name"]]
  rows.sort(key=lambda r:int(r["Dispatch_Id"]))
  return [(r["Kernel_Name"][:40], float(r["Counter_Value"])) for r in rows]
fe, wr = series("FETCH_SIZE"), series("WRITE_SIZE")
# 6 launches per case (1 warm + 5)
for i, c in enumerate(rates):
  f = [v for _, v in fe[6*i:6*i+6]]; w = [v for _, v in wr[6*i:6*i+6]]
  alg = c["elems"] * 2
  name = fe[6*i][0] if 6*i < len(fe) else "?"
  print("%-44s %-30s %.3f ms %.2f TB/s  fetch x%.2f  write x%.2f  %s" % (str(c["shape"])[:44], str(c["perm"])[:30], c["ms"], c["TBps"], 2*1024*sum(f)/len(f)/alg if f else 0, 1024*sum(w)/len(w)/alg if w else 0, name))
PY
