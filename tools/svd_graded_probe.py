"""f64 / f32 SVD on graded and rank-deficient matrices (what DMRG splits look like): python tools/svd_graded_probe.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
be = ta.get_hip_backend()
rng = np.random.default_rng(5)
for dt in (np.float64, np.float32):
  for n, kind in [(256, "graded2"), (256, "rank40"), (512, "graded1.2"), (256, "zero_rows"), (384, "dmrg_like")]:
    u, _ = np.linalg.qr(rng.standard_normal((n, n)))
    v, _ = np.linalg.qr(rng.standard_normal((n, n)))
    if kind == "graded2": s = 2.0 ** -np.arange(n)
    elif kind == "graded1.2": s = 1.2 ** -np.arange(n)
    elif kind == "rank40": s = np.concatenate([np.linspace(1, 0.1, 40), np.zeros(n - 40)])
    elif kind == "zero_rows": s = np.concatenate([np.ones(n // 2), np.zeros(n - n // 2)])
    else: s = np.exp(-np.arange(n) / 6.0)
    a = (u * s) @ v.T
    if kind == "zero_rows": a[::3] = 0.0
    a = a.astype(dt)
    x = be.convert_to_tensor(a)
    try:
      uu, ss, vv, _ = be.svd(x, 1)
      s_ref = np.linalg.svd(a.astype(np.float64), compute_uv=False)
      err = float(np.abs(np.asarray(ss) - s_ref).max() / s_ref[0])
      uh, vh = np.asarray(uu).astype(np.float64), np.asarray(vv).astype(np.float64)
      orth = max(float(np.abs(uh.T @ uh - np.eye(n)).max()), float(np.abs(vh @ vh.T - np.eye(n)).max()))
      rec = float(np.abs((uh * np.asarray(ss).astype(np.float64)) @ vh - a.astype(np.float64)).max() / s_ref[0])
      print(json.dumps({"dtype": np.dtype(dt).name, "n": n, "kind": kind, "sweeps": be.last_svd_sweeps, "s_err": err,
                        "orth": orth, "recon": rec}), flush=True)
    except Exception as exc:  # pylint: disable=broad-except
      print(json.dumps({"dtype": np.dtype(dt).name, "n": n, "kind": kind, "error": str(exc)[:120]}), flush=True)
