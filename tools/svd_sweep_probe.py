"""Convergence history and knob sweep of the block-Jacobi SVD at configs[2] size (GPU box only).

  python tools/svd_sweep_probe.py [--n 4096]

For each knob setting: seconds, sweeps, and (first arm, with TNH_SVD_TRACE) the largest normalised
off-diagonal per sweep; every arm's singular values are compared with the first arm's (float64 LAPACK
check of arm 0 when n <= 2048) so that a faster stop rule is only accepted if the values still agree to
1e-5 s0 and the kept vectors stay orthonormal to 1e-4."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--spectrum", default="gauss", choices=["gauss", "graded"])
a = ap.parse_args()
be = ta.get_hip_backend()
n, k = a.n, a.n // 16
if a.spectrum == "gauss":
  x = be.device_random((n, n), dtype=np.float32, seed=3, normal=True)
else:
  rng = np.random.default_rng(4)
  q1, _ = np.linalg.qr(rng.standard_normal((n, n)))
  q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
  x = be.convert_to_tensor(((q1 * 2.0 ** (-np.arange(n) / 32)) @ q2.T).astype(np.float32))

ARMS = [
    ("r1_classic", {"TNH_SVD_TRACE": "1", "TNH_SVD_STOP": "0", "TNH_SVD_SORT": "0", "TNH_SVD_GROUPS": "1"}),
    ("sort", {"TNH_SVD_TRACE": "1", "TNH_SVD_STOP": "0", "TNH_SVD_GROUPS": "1"}),
    ("sort_stop", {"TNH_SVD_GROUPS": "1"}),
    ("sort_stop_g2", {}),
    ("sort_stop_g3", {"TNH_SVD_GROUPS": "3"}),
    ("sort_stop_g4", {"TNH_SVD_GROUPS": "4"}),
    ("nosort_stop_g2", {"TNH_SVD_SORT": "0"}),
    ("sort_stop_g2_inner2", {"TNH_SVD_INNER": "2"}),
]
KNOBS = ("TNH_SVD_TRACE", "TNH_SVD_STOP", "TNH_SVD_INNER", "TNH_SVD_CROSS", "TNH_SVD_EIGNT", "TNH_SVD_SORT", "TNH_SVD_GROUPS")
ref = None
for name, env in ARMS:
  for kk in KNOBS:
    os.environ.pop(kk, None)
  os.environ.update(env)
  best = None
  for rep in range(2):
    if rep == 1:
      os.environ.pop("TNH_SVD_TRACE", None)
    be.synchronize()
    t0 = time.perf_counter()
    u, s, vh, rest = be.svd(x, 1, max_singular_values=k)
    be.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
  s_all = np.concatenate([np.asarray(s), np.asarray(rest)]).astype(np.float64)
  uu = np.asarray(u).astype(np.float64)
  vv = np.asarray(vh).astype(np.float64)
  rec = {"arm": name, "n": n, "sec": best, "sweeps": be.last_svd_sweeps,
         "orth_u": float(np.abs(uu.T @ uu - np.eye(k)).max()), "orth_vh": float(np.abs(vv @ vv.T - np.eye(k)).max())}
  if ref is None:
    ref = s_all
    if n <= 2048:
      sr = np.linalg.svd(np.asarray(x).astype(np.float64), compute_uv=False)
      rec["s_err_vs_lapack"] = float(np.abs(s_all - sr).max() / sr[0])
  else:
    rec["s_diff_vs_classic_over_s0"] = float(np.abs(s_all - ref).max() / ref[0])
  print(json.dumps(rec), flush=True)
