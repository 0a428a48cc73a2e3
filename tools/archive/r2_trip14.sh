#!/bin/bash
# Round-2 trip 14: streaming small x long GEMM kernel -- tests, A/B against the ragged tile kernels, sliced network.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "stream or ragged" > $OUT/pytest_stream.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_stream.log
echo "== A/B"
timeout 400 python tools/gemm_sweep.py --stream > $OUT/stream_ab.jsonl 2> $OUT/stream_ab.err; echo "rc=$?"; tail -3 $OUT/stream_ab.err
python - <<'PY'
import json
for l in open('gpurun_out/stream_ab.jsonl'):
  r = json.loads(l); print(r["m"], r["n"], r["k"], r["variant"], r["kernel"], round(r["ms"], 4), "ms", round(r["gbps"]), "GB/s")
PY
echo "== sliced network + forced-dist world 1"
for s in 1 0; do
  TNH_GEMM_STREAM=$s timeout 600 python bench.py --steps 3 --warmup 1 --svd-n 0 --mera-chi 0 --no-sweep --no-extras --no-cpu-baseline --no-verify > $OUT/sl_$s.json 2> $OUT/sl_$s.err; echo "rc=$?"
  python - "$OUT/sl_$s.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("sliced", round(r["sliced_network"]["seconds"], 4), round(r["sliced_network"]["tflops"]), "TF")
PY
done
TNH_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py --steps 3 --warmup 1 --svd-n 0 --mera-chi 0 --no-sweep --no-extras --no-cpu-baseline > $OUT/dist1.json 2> $OUT/dist1.err; echo "dist1 rc=$?"; tail -2 $OUT/dist1.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/dist1.json').read().strip().splitlines()[-1])
print("dist1 value", round(r["value"]), r["config"].get("communicator"), "verified", r.get("verified", {}).get("all_ok"))
PY
