#!/bin/bash
# Round-2 trip 8: what the driver runs at round end -- smoke, pytest -m gpu (incl. the reference drop-in), bench.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TN_REFERENCE_DIR=$PWD/_reference_scratch
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
echo "== bench (driver arguments)"
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print("value", r["value"], "ms/step", r["ms_per_step"], "verified.all_ok", r.get("verified", {}).get("all_ok"))
print("roofline", {k: r["roofline"][k] for k in ("achieved", "frac", "kernel", "observed_clock_mhz", "board_power_w", "frac_at_observed_clock", "traffic")})
for row in r.get("bond_sweep", []):
  print("   ", row["D"], row["layout"][:2], round(row["tflops"]), row["kernel"], row["permute_launches"])
for key in ("sliced_network", "mera", "svd"):
  print(key, json.dumps(r.get(key))[:400])
print("last stdout line is the JSON:", open('gpurun_out/bench.json').read().strip().splitlines()[-1][:20])
PY
