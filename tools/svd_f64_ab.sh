python -m pytest tests/test_gpu_linalg.py tests/test_gpu_mps.py -m gpu -q 2>&1 | tail -2
for e in 1 0; do echo "eig64=$e: $(TNH_SVD_EIG64=$e python tools/svd_probe.py --check 1 --sizes 2048,512 --reps 2 --dtype f64 2>&1 | tail -3 | tr '\n' ' ')"; done
