#!/bin/bash
# bench.py's gather_gemm leg alone (tools/gather_probe.py), with the device-side equality check.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t22; mkdir -p $O
timeout 150 python tools/gather_probe.py > $O/gather_gemm_rows.jsonl 2> $O/gather_probe.err; echo "probe rc=$?"; python - <<'PY'
import json
for line in open("gpurun_out/r4t22/gather_gemm_rows.jsonl"):
  r = json.loads(line)
  if "case" in r:
    print(r["case"], r["box"], {o: [round(r[o]["classic_us"]), round(r[o]["gather_us"]), r[o].get("max_abs_difference")] for o in ("small_first", "long_first")})
  else:
    print(r)
PY
tail -3 $O/gather_probe.err
