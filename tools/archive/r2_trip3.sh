#!/bin/bash
# Round-2 trip 3: full GPU suite, lowering-policy A/B over the bond sweep, the full bench line.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TN_REFERENCE_DIR=$PWD/_reference_scratch
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest_gpu.log
echo "== lowering A/B"
QUICK="--steps 3 --warmup 1 --svd-n 0 --rr-bond 0 --mera-chi 0 --no-extras --no-cpu-baseline --no-verify"
for arm in "permute:TNH_ABSORB_TRANSPOSES=0" "view2g:TNH_VIEW_KMAJOR_MAX_BYTES=2147483648" "kcontig_only:TNH_VIEW_KMAJOR_MAX_BYTES=0" "always:TNH_VIEW_KMAJOR_MAX_BYTES=1000000000000"; do
  name=${arm%%:*}; envs=${arm#*:}
  env $envs timeout 300 python bench.py $QUICK > $OUT/ab_$name.json 2> $OUT/ab_$name.err; echo "$name rc=$?"
  python - "$OUT/ab_$name.json" "$name" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "headline", round(r["value"]), "kernel", r["roofline"]["kernel"], "permutes/step", r["config"]["permute_launches_per_step"])
for row in r.get("bond_sweep", []):
  print("   ", row["D"], row["layout"][:2], round(row["tflops"]), row["kernel"], row["permute_launches"])
PY
done
echo "== full bench"
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -5 $OUT/bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print("value", r["value"], "ms/step", r["ms_per_step"])
print("roofline", {k: r["roofline"][k] for k in ("achieved", "frac", "kernel", "observed_clock_mhz", "board_power_w", "frac_at_observed_clock")})
print("verified", json.dumps(r.get("verified"), indent=1)[:4000])
for key in ("sliced_network", "mera", "mera_chi64", "mps_chain", "svd", "cpu_baseline"):
  print(key, json.dumps(r.get(key))[:900])
for row in r.get("helpers", []) if isinstance(r.get("helpers"), list) else [r.get("helpers")]:
  print("helper", row)
PY
