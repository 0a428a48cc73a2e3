#!/bin/bash
# as tools/raster_ab.sh over square / wide / tall grids and K = 1024 ... 8192 (the first value of an arm is a cold start)
K=auto:r0,auto:r1,auto:r0,auto:r1,auto:r0,auto:r1,auto,auto
for s in "32768 32768 1024" "32768 32768 1536" "32768 32768 2048" "32768 32768 3072" "32768 32768 4096" "32768 32768 6144" "32768 32768 8192" "65536 65536 2048" "65536 65536 4096" "65536 65536 8192" "16384 65536 2048" "16384 65536 4096" "65536 16384 2048" "65536 16384 4096" "65536 16384 8192" "20736 20736 1728" "16384 16384 4096" "8192 8192 4096"; do
  set -- $s
  python tools/view_probe.py --m $1 --n $2 --k $3 --va $1,$3,0,$3,1,0 --vb $2,$3,0,$3,1,0 --knobs $K --iters 4 | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r.setdefault(d['knob'],[]).append(d['tflops'])
print('$1 x $2 x $3', {k:[round(x) for x in v] for k,v in r.items()})"
done
