#!/bin/bash
# k-major operand in place vs NT on a copy: size dependence, and on zeros (no power cap)
set -u
O=gpurun_out/${1:-r3t21}
mkdir -p $O
rm -f $O/touch2.jsonl
timeout 300 python tools/kmajor_touch_probe.py --shapes 8192x8192x65536,8192x8192x262144,16384x16384x65536 >> $O/touch2.jsonl 2>> $O/touch.err
timeout 300 python tools/kmajor_touch_probe.py --zeros --shapes 8192x8192x65536,8192x8192x262144 >> $O/touch2.jsonl 2>> $O/touch.err
python - <<PY
import json
for l in open("$O/touch2.jsonl"):
  r = json.loads(l)
  print("%s zeros=%s  nt %.2f ms %.0f TF  permute %.2f ms  nt path %.0f TF | view %.2f ms %.0f TF  same=%s" % (
    (r["m"], r["n"], r["k"]), r["zeros"], r["nt_ms"], r["nt_tflops"], r["permute_ms"], r["nt_path_tflops"], r["view_ms"], r["view_tflops"], r["bit_identical"]))
PY
tail -3 $O/touch.err
