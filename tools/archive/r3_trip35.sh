#!/bin/bash
# brick permutes: workgroups per CU
set -u
O=gpurun_out/${1:-r3t35}
mkdir -p $O
for g in 16 4 8; do
  TNH_BRICK_GRID=$g timeout 300 python tools/permute_set_probe.py > $O/brick_g$g.jsonl 2>> $O/err.txt
done
python - <<PY
import json
rows = {g: [json.loads(l) for l in open("$O/brick_g%d.jsonl" % g)] for g in (16, 4, 8)}
for i, r in enumerate(rows[16]):
  print("%-46s %-30s %7.1f MB  g16 %.2f  g4 %.2f  g8 %.2f TB/s" % (str(r["shape"])[:46], str(r["perm"])[:30], r["elems"] * 2 / 1e6, r["TBps"], rows[4][i]["TBps"], rows[8][i]["TBps"]))
PY
