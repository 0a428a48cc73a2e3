"""QR / eigh timing probe (GPU box): python tools/qr_probe.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
be = ta.get_hip_backend()
for (m, n, dt) in [(4096, 4096, np.float32), (65536, 256, np.float32), (4096, 4096, np.float64), (1024, 1024, np.float32)]:
  x = be.device_random((m, n), dtype=dt, seed=1)
  be.qr(x, 1); be.synchronize()
  t0 = time.perf_counter(); q, r = be.qr(x, 1); be.synchronize(); t = time.perf_counter() - t0
  xh = np.asarray(x)
  t0 = time.perf_counter(); np.linalg.qr(xh); tc = time.perf_counter() - t0
  print(json.dumps({"op": "qr", "m": m, "n": n, "dtype": np.dtype(dt).name, "gpu_s": t, "numpy_s": tc,
                    "gflops": (2.0 * m * n * n - 2.0 * n ** 3 / 3) * 2 / t / 1e9}), flush=True)
for n in (1024, 2048):
  a = be.device_random((n, n), dtype=np.float32, seed=2)
  h = be.addition(a, be.transpose(a, (1, 0)))
  be.eigh(h); be.synchronize()
  t0 = time.perf_counter(); w, v = be.eigh(h); be.synchronize(); t = time.perf_counter() - t0
  hh = np.asarray(h)
  t0 = time.perf_counter(); np.linalg.eigh(hh); tc = time.perf_counter() - t0
  print(json.dumps({"op": "eigh", "n": n, "gpu_s": t, "numpy_s": tc, "svd_sweeps": be.last_svd_sweeps}), flush=True)
