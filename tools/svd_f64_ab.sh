python -m pytest tests/test_gpu_linalg.py tests/test_gpu_mps.py -m gpu -q 2>&1 | tail -2
for sc in 2 1; do echo "sched=$sc: $(TNH_SVD_SCHED=$sc python tools/svd_probe.py --check 1 --sizes 2048,512 --reps 2 --dtype f64 2>&1 | tail -3 | tr '\n' ' ')"; done
