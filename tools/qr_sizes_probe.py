"""QR timing over sizes (no NumPy leg): python tools/qr_sizes_probe.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
be = ta.get_hip_backend()
for (m, n) in [(4096, 4096), (2048, 2048), (1024, 1024), (2048, 512), (2560, 256), (65536, 256)]:
  x = be.device_random((m, n), dtype=np.float32, seed=1)
  be.qr(x, 1); be.synchronize()
  ts = []
  for _ in range(3):
    t0 = time.perf_counter(); q, r = be.qr(x, 1); be.synchronize(); ts.append(time.perf_counter() - t0)
  xh, qh, rh = np.asarray(x).astype(np.float64), np.asarray(q).astype(np.float64), np.asarray(r).astype(np.float64)
  err = np.abs(qh @ rh - xh).max() / np.abs(xh).max()
  orth = np.abs(qh.T @ qh - np.eye(n)).max()
  print(json.dumps({"m": m, "n": n, "sec": min(ts), "recon": err, "orth": orth}), flush=True)
