#!/bin/bash
set -u
O=gpurun_out/${1:-r3t15}
mkdir -p $O
timeout 300 python tools/energy_probe/run.py 1.5 > $O/energy.jsonl 2> $O/energy.err; echo "rc=$?"
python - <<PY
import json
for l in open("$O/energy.jsonl"):
  r=json.loads(l)
  if "tflops" in r: print("mode %d %-52s %7.0f TF %6.0f W %5.0f MHz %6.3f pJ/flop  %6.0f GB/s"%(r["mode"],r["what"],r["tflops"],r.get("power_mean_w") or 0,r.get("sclk_mean_mhz") or 0,r.get("pj_per_flop") or 0,r["global_GBps"]))
  else: print(r)
PY
tail -3 $O/energy.err
