#!/bin/bash
# Round-2 trip 17: brick permute with computed positions: tests, brick size arms, sliced network, helper bench.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== tests"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "permute or transpose or golden" > $OUT/pytest_perm.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_perm.log
echo "== permute set: brick payload 16 / 32 / 48 KiB"
for kb in 16 32 48; do TNH_BRICK_MAXKB=$kb python tools/permute_set_probe.py > $OUT/permute_set_kb$kb.jsonl 2>&1; done
python - <<'PY'
import json
arms = {kb: [json.loads(l) for l in open(f'gpurun_out/permute_set_kb{kb}.jsonl') if l.startswith('{')] for kb in (16, 32, 48)}
for i, x in enumerate(arms[16]):
  print(x["shape"], x["perm"], "  ".join(f"{kb}K {arms[kb][i]['ms']} ms {arms[kb][i]['TBps']} TB/s" for kb in (16, 32, 48)))
PY
echo "== sliced network"
for kb in 16 48; do
TNH_BRICK_MAXKB=$kb timeout 600 python bench.py --steps 3 --warmup 1 --svd-n 0 --mera-chi 0 --no-sweep --no-extras --no-cpu-baseline --no-verify > $OUT/sl_kb$kb.json 2> $OUT/sl.err; echo "rc=$?"
python - $kb <<'PY'
import json, sys
r = json.loads(open(f'gpurun_out/sl_kb{sys.argv[1]}.json').read().strip().splitlines()[-1])
print("maxkb", sys.argv[1], "sliced", round(r["sliced_network"]["seconds"], 4), round(r["sliced_network"]["tflops"]), "TF")
PY
done
