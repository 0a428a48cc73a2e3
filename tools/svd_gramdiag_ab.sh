for gd in 1 0; do
  echo "gramdiag=$gd: $(TNH_SVD_GRAMDIAG=$gd timeout 300 python tools/svd_probe.py --check 1 --sizes 4096,1024 --reps 2 2>&1 | tail -3 | tr '\n' ' ')"
done
TNH_SVD_GRAMDIAG=1 timeout 300 python tools/svd_sweep_probe.py 2>&1 | grep -v "^\[tnh" | head -6
