"""Round 6: the fast band reduction (tnh_svd_band_fast.inc) against the loop of rounds 3-5, same inputs.
  python tools/svd_fast_probe.py [--sizes 4096x4096,2048x2048,...] [--reps 5]
Per size and per TNH_SVDB_FAST in (1, 0): accuracy against float64 LAPACK (all values incl. s_rest, orthonormality,
the best rank-k approximation) and the best wall time of `reps` calls; plus graded / clustered spectra."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="4096x4096,2048x2048,1024x1024,512x512,3072x1024,4096x512,1000x600")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--spectra", type=int, default=1)
ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
a = ap.parse_args()
be = ta.get_hip_backend()
DT = np.float64 if a.dtype == "f64" else np.float32
TOL = (1e-12, 1e-10) if a.dtype == "f64" else (1e-5, 1e-4)


def run(x, k, fast, reps, check=True, tag=""):
  os.environ["TNH_SVDB_FAST"] = "1" if fast else "0"
  x = x.astype(DT)
  m, n = x.shape
  d = be.convert_to_tensor(x)
  best = None
  for _ in range(reps):
    be.synchronize()
    t0 = time.perf_counter()
    u, s, vh, rest = be.svd(d, 1, max_singular_values=k)
    be.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
  rec = {"shape": [m, n], "k": k, "fast_env": fast, "stage1": int(be.lib.tnh_svd_band_last_stage1()), "path": be.last_svd_path,
         "status": int(getattr(be, "last_svd_band_status", -1) or 0), "ms": best * 1e3, "tag": tag}
  if check:
    u, s, vh, rest = (np.asarray(t).astype(np.float64) for t in (u, s, vh, rest))
    x64 = x.astype(np.float64)
    ur, sr, vhr = np.linalg.svd(x64, full_matrices=False)
    s_all = np.concatenate([s, rest])
    rec["s_err"] = float(np.abs(s_all - sr).max() / sr[0])
    rec["orth_u"] = float(np.abs(u.T @ u - np.eye(k)).max())
    rec["orth_v"] = float(np.abs(vh @ vh.T - np.eye(k)).max())
    bestk = (ur[:, :k] * sr[:k]) @ vhr[:k]
    rec["recon_vs_best"] = float(np.linalg.norm((u * s) @ vh - bestk) / np.linalg.norm(x64))
    rec["s_kept_err"] = float(np.abs(s - sr[:k]).max() / sr[0])
    rec["ok"] = bool(rec["s_err"] < (1e-7 if a.dtype == "f64" else 1e-5) and rec["s_kept_err"] < TOL[0] * 10 and rec["orth_u"] < TOL[1] and rec["orth_v"] < TOL[1])
  print(json.dumps(rec), flush=True)
  return rec


rng = np.random.default_rng(5)
for sz in a.sizes.split(","):
  m, n = (int(v) for v in sz.split("x"))
  x = rng.standard_normal((m, n)).astype(np.float32)
  k = max(4, (min(m, n) // 16) // 4 * 4)
  for fast in (1, 0):
    run(x, k, fast, a.reps, check=bool(a.check) and max(m, n) <= 4096)

if a.spectra:
  def with_spectrum(m, n, spec):
    r = min(m, n)
    q1, _ = np.linalg.qr(rng.standard_normal((m, r)))
    q2, _ = np.linalg.qr(rng.standard_normal((n, r)))
    return ((q1 * spec) @ q2.T).astype(np.float32)
  n = 1024
  i = np.arange(n)
  for tag, spec in [("exp(-i/100)", np.exp(-i / 100.0)), ("2^(-i/32)", 2.0 ** (-i / 32.0)), ("1/(1+i)", 1.0 / (1.0 + i)),
                    ("pairs", np.repeat(np.linspace(2, 1, n // 2), 2)), ("half-rank", np.where(i < n // 2, np.linspace(2, 1, n), 0.0)),
                    ("uniform[1,2]", np.linspace(2, 1, n))]:
    x = with_spectrum(n, n, spec)
    for fast in (1, 0):
      run(x, 64, fast, 2, tag=tag)
  # an MPS-like two-site tensor: product of two random low-rank factors plus noise
  x = (rng.standard_normal((1024, 96)) @ rng.standard_normal((96, 1024)) + 1e-3 * rng.standard_normal((1024, 1024))).astype(np.float32)
  for fast in (1, 0):
    run(x, 96, fast, 2, tag="rank96+1e-3noise")
