#!/bin/bash
# chi = 64 / chi = 32 MERA-like passes (extents 64 or 32: b % 128 != 0 sends them past the 64 x 128 fast path)
for spec in "64,64,64,64,64 4,1,2,3,0" "64,64,64,64,64 1,2,3,4,0" "64,64,64,64,64 4,0,1,2,3" "64,64,64,64,64 2,3,4,0,1" "64,64,64,64,64 0,4,2,3,1" "4096,64,64 2,1,0" "64,4096,64 2,1,0" "64,64,4096 2,0,1" "192,64,192,64 3,2,1,0" "64,320,64 2,1,0"; do
  set -- $spec
  for knob in "TNH_BRICK_RAGGED=0" "TNH_BRICK_RAGGED=1"; do
    echo -n "$1 $2 $knob  "
    env $knob timeout 60 python tools/permute_one.py --shape $1 --perm $2 --iters 100 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms'],4), 'ms', round(r['TBps'],2), 'TB/s')"
  done
done
