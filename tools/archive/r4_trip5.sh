#!/bin/bash
# Round 4, trip 5: whole GPU suite (templated band SVD + f64 QR on the panel kernels), QR timings by dtype.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t5; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
timeout 300 python - > $O/qr.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np
import tensornetwork_amd as ta
be = ta.get_hip_backend()
for dt in (np.float32, np.float64):
  for (m, n) in ((4096, 4096), (65536, 256), (2048, 2048)):
    a = be.cast(be.device_random((m, n), dtype=np.float32, seed=m + n), dt)
    best = 1e9
    for rep in range(4):
      be.synchronize(); t0 = time.perf_counter()
      q, r = be.qr(a, 1)
      be.synchronize(); t = time.perf_counter() - t0
      if rep: best = min(best, t)
    err = ""
    if m * n <= 4096 * 4096:
      qh, rh, ah = np.asarray(q).astype(np.float64), np.asarray(r).astype(np.float64), np.asarray(a).astype(np.float64)
      err = "recon %.2e orth %.2e" % (np.max(np.abs(qh @ rh - ah)) / np.max(np.abs(ah)), np.max(np.abs(qh.T @ qh - np.eye(n))))
    print(np.dtype(dt).name, m, n, "%.2f ms" % (best * 1e3), err, flush=True)
PY
cat $O/qr.txt
