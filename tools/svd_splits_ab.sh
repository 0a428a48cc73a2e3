for sp in 4 8 16; do
  echo "splits=$sp: $(TNH_SVD_SPLITS=$sp timeout 300 python tools/svd_probe.py --check 0 --sizes 4096,1024 --reps 2 2>&1 | tail -3 | tr '\n' ' ')"
done
