#!/bin/bash
# [K][N] -> [N][K]-like passes of 2-byte tensors whose extents the 64 x 128 tiled fast path cannot tile: the scalar
# 64 x 64 fallback (TNH_BRICK_RAGGED=0) against the brick kernel (=1), bricks of 32 and 40 KiB
for spec in "96,96,96,96 1,3,2,0" "96,96,96,96 3,2,1,0" "96,96,96,96 2,3,0,1" "160,160,160,160 1,3,2,0" "72,72,72,72 1,3,2,0" "192,192,96,96 2,3,0,1" "96,96,192,192 2,3,0,1" "9216,9216 1,0" "100,100,100,100 1,3,2,0" "12,12,12,12,12,12,12 6,4,2,0,1,3,5"; do
  set -- $spec
  for knob in "TNH_BRICK_RAGGED=0" "TNH_BRICK_RAGGED=1" "TNH_BRICK_RAGGED=1 TNH_BRICK_MAXKB=40"; do
    echo -n "$1 $2 $knob  "
    env $knob timeout 60 python tools/permute_one.py --shape $1 --perm $2 --iters 300 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms'],4), 'ms', round(r['TBps'],2), 'TB/s')"
  done
done
