python -m pytest tests/test_gpu_linalg.py -m gpu -q 2>&1 | tail -2
python tools/svd_probe.py --check 1 --sizes 2048,512 --reps 2 --dtype f64 2>&1 | tail -3
python tools/qr_sizes_probe.py 2>&1 | tail -6
