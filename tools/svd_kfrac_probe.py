"""Band vs Jacobi by kept fraction k / r (the DMRG two-site split keeps k = r / 2): ms per call, best of 3."""
import json, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
be = ta.get_hip_backend()
for dt in (np.float32, np.float64):
  for n in (512, 1024, 2048):
    a = be.cast(be.device_random((n, n), dtype=np.float32, seed=n), dt)
    for frac in (16, 4, 2, 1):
      k = n // frac
      row = {"dtype": np.dtype(dt).name, "n": n, "k": k}
      for band in (True, False):
        be.svd_band = band
        best = 1e9
        for rep in range(4):
          be.synchronize(); t0 = time.perf_counter()
          out = be.svd(a, 1, max_singular_values=k)
          be.synchronize(); t = time.perf_counter() - t0
          if rep: best = min(best, t)
        row["band_ms" if band else "jacobi_ms"] = round(best * 1e3, 2)
        row["path_" + ("band" if band else "jacobi")] = be.last_svd_path
      be.svd_band = True
      print(json.dumps(row), flush=True)
