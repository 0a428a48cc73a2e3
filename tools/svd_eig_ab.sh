for b in 1 0; do for nt in 1024 512; do
  echo "bcast=$b eig_nt=$nt: $(TNH_SVD_TRACE=1 TNH_SVD_BCAST=$b TNH_SVD_EIGNT=$nt python tools/svd_probe.py --check 0 --sizes 4096 --reps 1 2>&1 | grep -v 'sweep [0-9]:\|sweep 1[0-6]' | tail -3 | tr '\n' ' ')"
done; done
