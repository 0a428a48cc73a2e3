#!/bin/bash
# A/B of the non-temporal epilogue stores of the ping-pong GEMM (results >= 64 MiB; knob :n0 = plain stores), arms alternating
K=auto:n0,auto,auto:n0,auto,auto:n0,auto,auto:n0,auto
for s in "32768 32768 1024" "32768 32768 2048" "32768 32768 4096" "65536 65536 4096" "1048576 32768 1024" "16384 16384 16384" "65536 65536 8192" "20736 20736 1728"; do
  set -- $s
  python tools/view_probe.py --m $1 --n $2 --k $3 --va $1,$3,0,$3,1,0 --vb $2,$3,0,$3,1,0 --knobs $K --iters 4 | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r.setdefault(d['knob'],[]).append(d['tflops'])
print('$1 x $2 x $3', {k:[round(x) for x in v] for k,v in r.items()})"
done
