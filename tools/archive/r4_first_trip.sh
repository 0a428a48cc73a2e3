#!/bin/bash
# First GPU trip of the next round (round 3 ended without GPU minutes for a last full run on the final tree):
#   1. the whole GPU suite, 2. the default bench line, 3. row-padded results A/B on the MERA chi = 32 layer
#   (HipBackend.pad_results: written and CPU-validated in round 3, never measured), 4. the same on the D = 512 row.
# usage: /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/r4_first_trip.sh'
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
for pad in 0 1 0 1; do
  TNH_PAD_RESULTS=$pad timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sweep --no-extras --mera-chi 32 \
    --svd-n 0 --rr-bond 0 --bond 64 > $O/mera_pad$pad.json 2>> $O/mera.err
  python - <<PY
import json
r = json.loads(open("$O/mera_pad$pad.json").read().strip().splitlines()[-1])
print("pad_results=$pad  MERA chi=32: %.4f s  %.0f TFLOP/s  permutes %s  verified %s" % (
  r["mera"]["seconds"], r["mera"]["tflops"], r["mera"]["permute_launches"], r["verified"].get("all_ok")))
PY
done
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4t1/bench.json").read().strip().splitlines()[-1])
print("value", r["value"], "frac", r["roofline"]["frac"], "all_ok", r["verified"]["all_ok"])
for row in r.get("bond_sweep", []):
  print(row["D"], row["layout"][:2], round(row["tflops"]), row["permute_launches"])
print("svd", r["svd"]["seconds"], r["svd"].get("samples_ms"), "mera", r["mera"]["tflops"], "rr", r["sliced_network"]["tflops"])
for h in r.get("helpers", []):
  print("%-70s %.0f GB/s" % (h["op"][:70], h["gbps"]))
PY
