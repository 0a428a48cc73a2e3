python tools/svd_graded_probe.py 2>&1 | tail -10 | cut -c1-200
python -m pytest tests/test_gpu_linalg.py tests/test_gpu_mps.py -m gpu -q 2>&1 | tail -2
python tools/svd_probe.py --check 0 --sizes 4096 --reps 2 2>&1 | tail -2
