"""tensordot throughput per dtype through the backend (whole path: lowering + kernels).
  python tools/dtype_probe.py [--n 4096]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=4096); a = ap.parse_args()
be = ta.get_hip_backend(); n = a.n
for name, dt, mult in [("bf16", ta.bfloat16, 2), ("f16", np.float16, 2), ("f32", np.float32, 2), ("f64", np.float64, 2),
                       ("c64", np.complex64, 8), ("c128", np.complex128, 8)]:
  x = be.device_random((n, n), dtype=dt, seed=1); y = be.device_random((n, n), dtype=dt, seed=2)
  be.tensordot(x, y, 1); be.synchronize()
  reps = 5
  t0 = time.perf_counter()
  for _ in range(reps): out = be.tensordot(x, y, 1)
  be.synchronize(); t = (time.perf_counter() - t0) / reps
  print(json.dumps({"dtype": name, "n": n, "ms": t * 1e3, "tflops": mult * n**3 / t / 1e12,
                    "kernel": be.lib.tnh_gemm_last_kernel().decode()}), flush=True)
