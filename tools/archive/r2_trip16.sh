#!/bin/bash
# Round-2 trip 16: 16-byte brick permutes -- tests, the D=12 network's permute set, sliced network.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== tests"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "permute or transpose or golden" > $OUT/pytest_perm.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_perm.log
echo "== permute set (vec16 on / off)"
python tools/permute_set_probe.py > $OUT/permute_set_v16.jsonl 2>&1
TNH_BRICK_VEC16=0 python tools/permute_set_probe.py > $OUT/permute_set_v4.jsonl 2>&1
python - <<'PY'
import json
a = [json.loads(l) for l in open('gpurun_out/permute_set_v16.jsonl') if l.startswith('{')]
b = [json.loads(l) for l in open('gpurun_out/permute_set_v4.jsonl') if l.startswith('{')]
for x, y in zip(a, b):
  print(x["shape"], x["perm"], "vec16", x["ms"], "ms", x["TBps"], "TB/s   dword", y["ms"], "ms", y["TBps"], "TB/s")
PY
echo "== sliced network"
timeout 600 python bench.py --steps 3 --warmup 1 --svd-n 0 --mera-chi 0 --no-sweep --no-extras --no-cpu-baseline --no-verify > $OUT/sl.json 2> $OUT/sl.err; echo "rc=$?"
python - <<'PY'
import json
r = json.loads(open('gpurun_out/sl.json').read().strip().splitlines()[-1])
print("sliced", round(r["sliced_network"]["seconds"], 4), round(r["sliced_network"]["tflops"]), "TF")
PY
