#!/bin/bash
# as tools/raster_ab.sh: the group size of the M-grouped order (:r2 = 4 tile rows, :r0 = 8, :r3 = 16, :r4 = 32) and the
# super-tiles (:r1) by K; 'auto' = pick_raster
K=auto:r0,auto:r1,auto:r2,auto:r3,auto:r4,auto:r0,auto:r1,auto:r2,auto:r3,auto:r4,auto:r0,auto:r1,auto:r2,auto:r3,auto:r4,auto,auto
for s in "32768 32768 1024" "65536 65536 1024" "131072 131072 512" "32768 1048576 1024" "32768 32768 1536" "262144 32768 512" "262144 32768 1536" "1048576 32768 1024" "32768 32768 2048" "32768 32768 4096" "65536 65536 4096" "262144 4096 4096" "524288 16384 2048"; do
  set -- $s
  python tools/view_probe.py --m $1 --n $2 --k $3 --va $1,$3,0,$3,1,0 --vb $2,$3,0,$3,1,0 --knobs $K --iters 4 | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r.setdefault(d['knob'],[]).append(d['tflops'])
print('$1 x $2 x $3', {k:[round(x) for x in v] for k,v in r.items()})"
done
