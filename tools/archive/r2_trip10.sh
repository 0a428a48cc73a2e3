#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== f32 v2 tests + gemm tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "gemm or tensordot or complex" > $OUT/pytest_k.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest_k.log
echo "== f32 probe (v2)"
timeout 300 python tools/f32_gemm_probe.py > $OUT/f32_v2.jsonl 2>&1; echo "rc=$?"
echo "== f32 probe (old kernel)"
TNH_F32_V2=0 timeout 300 python tools/f32_gemm_probe.py > $OUT/f32_v1.jsonl 2>&1; echo "rc=$?"
python - <<'PY'
import json
def load(p):
  out = {}
  for l in open(p):
    try: r = json.loads(l)
    except ValueError: print(l.strip()); continue
    out[(r["dtype"], r["tA"], r["tB"], r["m"], r["n"], r["k"])] = r
  return out
a, b = load('gpurun_out/f32_v2.jsonl'), load('gpurun_out/f32_v1.jsonl')
for key in a:
  print(key, "v2", round(a[key]["tflops"], 1), a[key]["kernel"], "| old", round(b.get(key, {}).get("tflops", 0), 1), b.get(key, {}).get("kernel"))
PY
echo "== tail probe"
timeout 300 python tools/tail_probe.py > $OUT/tail_probe.jsonl 2>&1; echo "rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/tail_probe.jsonl'):
  try: r = json.loads(l)
  except ValueError: print(l.strip()); continue
  print(r["m"], r["n"], r["k"], r["knob"], r["rep"], round(r["ms"], 4), round(r["tflops"]), r["kernel"])
PY
