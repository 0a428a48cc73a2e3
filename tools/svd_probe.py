"""SVD probe on the GPU box: parity of HipBackend.svd against numpy.linalg.svd on a
few shapes, then wall-clock of the factorisation for square sizes.
  python tools/svd_probe.py [--check 1] [--sizes 1024,2048,4096] [--reps 1]
Knobs (env, read by libtnhip): TNH_SVD_BLOCK=0 (old 2-row kernel), TNH_SVD_INNER, TNH_SVD_INNER0."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta

ap = argparse.ArgumentParser()
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--sizes", default="1024,2048,4096")
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--verify-big", type=int, default=1, help="compare the largest size against numpy too")
a = ap.parse_args()
be = ta.get_hip_backend()


def check(shape, dtype=np.float32, seed=0, spectrum=None):
  rng = np.random.default_rng(seed)
  if spectrum is None:
    x = rng.standard_normal(shape).astype(dtype)
  else:
    m, n = shape
    r = min(m, n)
    q1, _ = np.linalg.qr(rng.standard_normal((m, r)))
    q2, _ = np.linalg.qr(rng.standard_normal((n, r)))
    x = ((q1 * spectrum(r)) @ q2.T).astype(dtype)
  k = max(1, min(shape) // 4)
  t0 = time.perf_counter()
  u, s, vh, rest = be.svd(be.convert_to_tensor(x), 1, max_singular_values=k)
  be.synchronize()
  dt = time.perf_counter() - t0
  u, s, vh, rest = (np.asarray(t).astype(np.float64) for t in (u, s, vh, rest))
  sr = np.linalg.svd(x.astype(np.float64), compute_uv=False)
  s_all = np.concatenate([s, rest])
  s_err = np.abs(s_all - sr).max() / sr[0]
  ou = np.abs(u.T @ u - np.eye(k)).max()
  ov = np.abs(vh @ vh.T - np.eye(k)).max()
  ur, srr, vhr = np.linalg.svd(x.astype(np.float64), full_matrices=False)
  best = (ur[:, :k] * srr[:k]) @ vhr[:k]
  rec = np.linalg.norm((u * s) @ vh - best) / np.linalg.norm(x)
  ok = s_err < 1e-5 and ou < 1e-4 and ov < 1e-4 and rec < 1e-3
  print(json.dumps({"shape": shape, "k": k, "s_err": s_err, "orth_u": ou, "orth_v": ov, "recon": rec,
                    "sweeps": be.last_svd_sweeps, "sec": dt, "ok": bool(ok)}), flush=True)
  return ok


allok = True
if a.check:
  for shape in [(65, 65), (130, 131), (96, 300), (300, 96), (257, 129), (512, 512), (700, 1000)]:
    allok &= check(shape)
  allok &= check((512, 512), spectrum=lambda r: 2.0 ** (-np.arange(r) / 32))
  allok &= check((384, 640), spectrum=lambda r: np.where(np.arange(r) < r // 2, 1.0, 0.0) * np.linspace(1, 2, r))

for n in [int(v) for v in a.sizes.split(",") if v]:
  x = be.device_random((n, n), dtype=np.float64 if a.dtype == 'f64' else np.float32, seed=3, normal=True)
  k = n // 16
  best = None
  for _ in range(a.reps + 1):
    be.synchronize()
    t0 = time.perf_counter()
    u, s, vh, rest = be.svd(x, 1, max_singular_values=k)
    be.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
  nbytes = x.itemsize * (n * n + n * k + n + k * n)
  rec = {"n": n, "k": k, "sec": best, "gbps": nbytes / best / 1e9, "sweeps": be.last_svd_sweeps}
  if a.verify_big and n <= 2048:
    xs = np.asarray(x).astype(np.float64)
    sr = np.linalg.svd(xs, compute_uv=False)
    s_all = np.concatenate([np.asarray(s), np.asarray(rest)]).astype(np.float64)
    rec["s_err"] = float(np.abs(s_all - sr).max() / sr[0])
    uu = np.asarray(u).astype(np.float64)
    rec["orth_u"] = float(np.abs(uu.T @ uu - np.eye(k)).max())
  print(json.dumps(rec), flush=True)
print("ALLOK" if allok else "SOMEFAIL")
