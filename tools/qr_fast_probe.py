"""Round 6: the fused-kernel Householder QR (qr_panels_fast) against the five-launch loop (TNH_SVDB_FAST=0), same inputs:
|Q R - A|, |Q^T Q - I|, R against np.linalg.qr (same reflector signs), best wall time of `reps` calls.
  python tools/qr_fast_probe.py [--dtype f32|f64]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="f32"); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--sizes", default="4096x4096,2048x2048,1024x512,65536x256,8192x1024,1000x512")
a = ap.parse_args()
be = ta.get_hip_backend()
DT = np.float64 if a.dtype == "f64" else np.float32
rng = np.random.default_rng(9)
for sz in a.sizes.split(","):
  m, n = (int(v) for v in sz.split("x"))
  x = rng.standard_normal((m, n)).astype(DT)
  d = be.convert_to_tensor(x)
  for fast in (1, 0):
    os.environ["TNH_SVDB_FAST"] = str(fast)
    best = None
    for _ in range(a.reps):
      be.synchronize(); t0 = time.perf_counter()
      q, r = be.qr(d, 1, False)
      be.synchronize(); dt = time.perf_counter() - t0
      best = dt if best is None else min(best, dt)
    rec = {"shape": [m, n], "dtype": a.dtype, "fast_env": fast, "ms": best * 1e3}
    if m * n <= 4096 * 4096:
      qh, rh = np.asarray(q).astype(np.float64), np.asarray(r).astype(np.float64)
      x64 = x.astype(np.float64)
      rec["resid"] = float(np.abs(qh @ rh - x64).max() / np.abs(x64).max())
      rec["orth"] = float(np.abs(qh.T @ qh - np.eye(n)).max())
      if m <= 4096:
        ro = np.linalg.qr(x64, mode="r")
        rec["r_vs_lapack"] = float(np.abs(rh - ro).max() / np.abs(ro).max())
    print(json.dumps(rec), flush=True)
