"""Times the band SVD (K7b) phase by phase and checks it against LAPACK: python tools/svd_band_probe.py [n] [k] [kind]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
  k = int(sys.argv[2]) if len(sys.argv) > 2 else n // 16
  kind = sys.argv[3] if len(sys.argv) > 3 else "gauss"
  check = "--no-check" not in sys.argv
  be = ta.get_hip_backend()
  lib = be.lib
  rng = np.random.default_rng(1)
  if kind == "gauss":
    a = rng.standard_normal((n, n)).astype(np.float32)
  else:
    qu, _ = np.linalg.qr(rng.standard_normal((n, n)))
    qv, _ = np.linalg.qr(rng.standard_normal((n, n)))
    a = ((qu * 2.0 ** (-np.arange(n) / 32.0)) @ qv.T).astype(np.float32)
  da = be.convert_to_tensor(a)
  nbytes = ctypes.c_size_t(0)
  _lib.check(lib.tnh_svd_band_work_bytes(_lib.F32, n, n, k, ctypes.byref(nbytes)))
  work = DeviceTensor.empty((nbytes.value // 8 + 1,), _lib.F64)
  s_all = DeviceTensor.empty((n,), _lib.F32)
  u = DeviceTensor.empty((n, k), _lib.F32)
  vh = DeviceTensor.empty((k, n), _lib.F32)
  st1, st2 = ctypes.c_int(0), ctypes.c_int(0)
  rec = {"n": n, "k": k, "kind": kind, "work_MB": nbytes.value / 1e6,
         "env": {e: os.environ[e] for e in os.environ if e.startswith("TNH_SVDB")}}
  for rep in range(3):
    be.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.tnh_svd_band_factor(_lib.F32, n, n, ctypes.c_void_p(da.ptr), ctypes.c_void_p(s_all.ptr),
                                       ctypes.c_void_p(work.ptr), k, ctypes.byref(st1)), "factor")
    t1 = time.perf_counter()
    _lib.check(lib.tnh_svd_band_vectors(_lib.F32, n, n, ctypes.c_void_p(work.ptr), k, k, ctypes.c_void_p(u.ptr),
                                        ctypes.c_void_p(vh.ptr), None, ctypes.byref(st2)), "vectors")
    t2 = time.perf_counter()
    rec[f"rep{rep}"] = {"factor_ms": (t1 - t0) * 1e3, "vectors_ms": (t2 - t1) * 1e3, "total_ms": (t2 - t0) * 1e3,
                        "status": [st1.value, st2.value]}
  # whole call through the backend (truncation rule, host logic)
  x = be.svd(da, 1, max_singular_values=k)
  be.synchronize()
  t0 = time.perf_counter()
  x = be.svd(da, 1, max_singular_values=k)
  be.synchronize()
  rec["backend_svd_ms"] = (time.perf_counter() - t0) * 1e3
  rec["path"] = be.last_svd_path
  if check:
    sr = np.linalg.svd(a.astype(np.float64), compute_uv=False)
    s = np.asarray(s_all, dtype=np.float64)
    uu, vv = np.asarray(u, dtype=np.float64), np.asarray(vh, dtype=np.float64)
    rec["s_err_over_s0"] = float(np.max(np.abs(s - sr)) / sr[0])
    rec["orth_u"] = float(np.max(np.abs(uu.T @ uu - np.eye(k))))
    rec["orth_v"] = float(np.max(np.abs(vv @ vv.T - np.eye(k))))
    rec["resid_over_s0"] = float(np.max(np.linalg.norm(a.astype(np.float64) @ vv.T - uu * s[:k], axis=0)) / sr[0])
  print(json.dumps(rec))


if __name__ == "__main__":
  main()
