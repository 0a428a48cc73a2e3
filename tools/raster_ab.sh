#!/bin/bash
# A/B of the two tile orders of the 256 x 256 ping-pong GEMM (knobs :r0 = per-XCD M-grouped, :r1 = 16 x 16 super-tiles shared
# by the XCDs) on tall / wide / square results with short contractions, arms alternating four times; "auto" = pick_raster.
#   gpurun -- 'bash tools/raster_ab.sh'
K=auto:r0,auto:r1,auto:r0,auto:r1,auto:r0,auto:r1,auto:r0,auto:r1,auto,auto
for s in "1048576 32768 1024" "262144 32768 1024" "32768 1048576 1024" "524288 16384 2048" "262144 8192 2048" "262144 4096 4096" "131072 131072 512" "65536 65536 4096"; do
  set -- $s
  python tools/view_probe.py --m $1 --n $2 --k $3 --va $1,$3,0,$3,1,0 --vb $2,$3,0,$3,1,0 --knobs $K --iters 3 | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r.setdefault(d['knob'],[]).append(d['tflops'])
print('$1 x $2 x $3', {k:[round(x) for x in v] for k,v in r.items()})"
done
