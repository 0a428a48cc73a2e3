#!/bin/bash
# Round 4, trip 1: the whole GPU suite on the new tree, the driver's bench command (compact line + detail file),
# row-padded results A/B on the MERA chi = 32 layer (never measured in round 3).
# usage: /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r4_trip1.sh'
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --mera64-budget 30 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
tail -c 4200 $O/bench.out; echo; wc -c $O/bench.out
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
for pad in 0 1; do
  TNH_PAD_RESULTS=$pad timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sweep --no-extras --mera-chi 32 \
    --svd-n 0 --rr-bond 0 --bond 64 > $O/mera_pad$pad.out 2>> $O/mera.err
  cp gpurun_out/bench_detail.json $O/mera_pad${pad}_detail.json
  python - <<PY
import json
r = json.loads(open("$O/mera_pad${pad}_detail.json").read())
print("pad_results=$pad  MERA chi=32: %.4f s  %.0f TFLOP/s  permutes %s  verified %s" % (
  r["mera"]["seconds"], r["mera"]["tflops"], r["mera"]["permute_launches"], r["verified"].get("all_ok")))
PY
done
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4t1/bench_detail.json").read())
print("value", r["value"], "frac", r["roofline"]["frac"], "all_ok", r["verified"]["all_ok"])
print({k: v.get("ok") for k, v in r["verified"].items() if isinstance(v, dict)})
for row in r.get("bond_sweep", []):
  print(row["D"], row["layout"][:2], round(row["tflops"]), row["permute_launches"])
for row in r["svd"]["sweep"]:
  if row.get("mode") or (row["input"] == "gauss" and row["order"] == "natural"):
    print("svd", row.get("dtype", "f32"), row["n"], row["k"], row.get("mode"), row["path"], "%.2f ms" % (row["seconds"] * 1e3), row.get("check", {}).get("ok"), row.get("check"))
print(r["svd"].get("call_shapes_error"))
m = r["mera_chi64"]; print("mera64", {k: m.get(k) for k in ("measured_slices", "measured_seconds", "measured_tflops", "layer_seconds_1gpu_extrapolated", "error")})
print("mera", r["mera"].get("tflops"), "rr", r["sliced_network"].get("tflops"))
for h in r.get("helpers", []):
  print("%-78s %6.0f GB/s  x%d" % (h["op"][:78], h["gbps"], h.get("buffers", 0)))
PY
