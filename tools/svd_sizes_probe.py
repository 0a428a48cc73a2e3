"""Band SVD timings by dtype / size / call shape (best of 3 after a warm-up), with the path taken and the errors vs LAPACK
at the smaller sizes.  One JSON line per case."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import tensornetwork_amd as ta  # noqa: E402

be = ta.get_hip_backend()


def run(tag, a_dev, check=None, **kw):
  best = 1e9
  for rep in range(4):
    be.synchronize()
    t0 = time.perf_counter()
    out = be.svd(a_dev, 1, **kw)
    be.synchronize()
    t = time.perf_counter() - t0
    if rep:
      best = min(best, t)
  rec = {"case": tag, "ms": best * 1e3, "path": be.last_svd_path, "status": be.last_svd_band_status, "k": int(out[1].shape[0])}
  if check is not None:
    a = check
    sr = np.linalg.svd(a, compute_uv=False)
    k = rec["k"]
    sk, rest = np.real(np.asarray(out[1])).astype(np.float64), np.real(np.asarray(out[3])).astype(np.float64)
    u, vh = np.asarray(out[0]).astype(a.dtype), np.asarray(out[2]).astype(a.dtype)
    rec.update({"kept_err": float(np.max(np.abs(sk - sr[:k])) / sr[0]),
                "rest_err": float(np.max(np.abs(rest - sr[k:])) / sr[0]) if rest.size else 0.0,
                "orth_u": float(np.max(np.abs(u.conj().T @ u - np.eye(k)))), "orth_v": float(np.max(np.abs(vh @ vh.conj().T - np.eye(k)))),
                "resid": float(np.max(np.linalg.norm(a @ vh.conj().T - u * sk, axis=0)) / sr[0])})
  print(json.dumps(rec), flush=True)


rng = np.random.default_rng(0)
for n in (1024, 2048, 4096):
  a32 = be.device_random((n, n), dtype=np.float32, seed=n)
  host = np.asarray(a32).astype(np.float64) if n <= 2048 else None
  run(f"f32 {n} keep {n // 16}", a32, host, max_singular_values=n // 16)
  a64 = be.cast(a32, np.float64)
  run(f"f64 {n} keep {n // 16}", a64, host, max_singular_values=n // 16)
  be.svd_band_f64 = False
  run(f"f64 {n} keep {n // 16} (jacobi)", a64, None, max_singular_values=n // 16)
  be.svd_band_f64 = True
  del a32, a64
a = be.device_random((2048, 2048), dtype=np.float64, seed=5)
run("f64 2048 full", a, np.asarray(a))
c = be.convert_to_tensor((rng.standard_normal((1024, 1024)) + 1j * rng.standard_normal((1024, 1024))).astype(np.complex64))
run("c64 1024 keep 64", c, np.asarray(c).astype(np.complex128), max_singular_values=64)
c = be.convert_to_tensor(rng.standard_normal((1024, 1024)) + 1j * rng.standard_normal((1024, 1024)))
run("c128 1024 keep 64", c, np.asarray(c), max_singular_values=64)
a = be.device_random((2008, 2008), dtype=np.float32, seed=6)
run("f32 2008 keep 125 (padded)", a, np.asarray(a).astype(np.float64), max_singular_values=125)
