for spec in "12,12,12,12,12,12,12,1,12 0,3,7,8,4,2,5,6,1" "12,12,12,12,12,1,12,12,12 0,4,5,6,1,2,7,8,3" "12,12,12,12,12,12,12 0,1,2,4,6,5,3" "12,1,12,12,12,12,12,12 1,5,2,7,3,6,0,4" "16,16,16,16,16,16,16 2,1,3,4,5,6,0" "8,8,8,8,8,8,8,8,8 0,4,5,6,1,2,7,8,3"; do
  set -- $spec
  for arm in "default" "512 512" "1024 1024" "2048 1024" "1024 256" "256 1024"; do
    if [ "$arm" = "default" ]; then
      r=$(TNH_BRICK_MAXKB=48 python tools/permute_one.py --shape $1 --perm $2 --iters 5 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms'],4), round(r['TBps'],2))")
    else
      set -- $1 $2 $arm
      r=$(TNH_BRICK_MAXKB=48 TNH_BRICK_TA=$3 TNH_BRICK_TB=$4 python tools/permute_one.py --shape $1 --perm $2 --iters 5 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms'],4), round(r['TBps'],2))")
    fi
    echo "$1 $2 [$arm] $r"
  done
done
