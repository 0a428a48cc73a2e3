#!/bin/bash
# tile raster A/B at the headline / north-star shapes
set -u
O=gpurun_out/${1:-r3t14}
mkdir -p $O
for shape in "65536 65536 65536 2" "8192 8192 262144 3" "36864 36864 36864 2" "16384 16384 16384 4" "4096 4096 4096 20"; do
  set -- $shape
  for v in auto bf16_256pp:r0 bf16_256pp:r2; do
    timeout 200 python tools/gemm_one.py --variant $v --m $1 --n $2 --k $3 --iters $4 --fill normal >> $O/raster.jsonl 2>> $O/err.txt
  done
done
python - <<PY
import json
for l in open("$O/raster.jsonl"):
  r=json.loads(l); print("%6d %6d %7d %-16s %8.1f TF"%(r["m"],r["n"],r["k"],r["variant"],r["tflops"]))
PY
tail -3 $O/err.txt
