#!/bin/bash
# k-major in place: tile raster A/B
set -u
O=$PWD/gpurun_out/${1:-r3t22}
mkdir -p $O
rm -f $O/raster.jsonl
for v in auto:r0 auto:r1; do
  timeout 300 python tools/kmajor_touch_probe.py --zeros --variant $v --shapes 8192x8192x262144 >> $O/raster.jsonl 2>> $O/span.err
  timeout 300 python tools/kmajor_touch_probe.py --variant $v --shapes 8192x8192x262144 >> $O/raster.jsonl 2>> $O/span.err
done
python - <<PY
import json
for l in open("$O/raster.jsonl"):
  r = json.loads(l)
  print("%s zeros=%s %s  nt %.2f ms %.0f TF | view %.2f ms %.0f TF  same=%s" % (
    (r["m"], r["n"], r["k"]), r["zeros"], r["variant"], r["nt_ms"], r["nt_tflops"], r["view_ms"], r["view_tflops"], r["bit_identical"]))
PY
tail -3 $O/span.err
