#!/bin/bash
# Round-2 trip 9: lowering policy re-check (MERA chi=32, bond sweep twice), skinny ragged tile A/B.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== kernel tests touched"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "view or ragged or gemm" > $OUT/pytest_k.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_k.log
echo "== skinny A/B"
timeout 300 python tools/gemm_sweep.py --skinny --variants x > $OUT/skinny.jsonl 2> $OUT/skinny.err; echo "rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/skinny.jsonl'):
  r = json.loads(l); print(r["m"], r["n"], r["k"], r["variant"], r["kernel"], round(r["ms"], 4), "ms", round(r["gbps"]), "GB/s")
PY
tail -3 $OUT/skinny.err
echo "== bench: mera + sweep x2"
for i in 1 2; do
  timeout 600 python bench.py --steps 3 --warmup 1 --svd-n 0 --rr-bond 12 --no-extras --no-cpu-baseline --no-verify > $OUT/b$i.json 2> $OUT/b$i.err; echo "rc=$?"
  python - "$OUT/b$i.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(r["value"]), "mera", round(r["mera"]["tflops"]), r["mera"]["permute_launches"], "sliced", round(r["sliced_network"]["seconds"], 3))
for row in r.get("bond_sweep", []):
  print("   ", row["D"], row["layout"][:2], round(row["tflops"]), row["kernel"], row["permute_launches"])
PY
done
