#!/bin/bash
# Round 4, trip 3: whole GPU suite on the templated band SVD (f32 + f64), probe of the complex128 case.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t3; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
timeout 300 python - > $O/c128.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np
import tensornetwork_amd as ta
be = ta.get_hip_backend()
rng = np.random.default_rng(0)
for n, k in ((1024, 64), (2048, 128)):
  c = be.convert_to_tensor(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
  for rep in range(3):
    be.synchronize(); t0 = time.perf_counter()
    out = be.svd(c, 1, max_singular_values=k)
    be.synchronize(); t = time.perf_counter() - t0
  print("c128", n, k, "%.1f ms" % (t * 1e3), be.last_svd_path, be.last_svd_band_status)
PY
cat $O/c128.txt
