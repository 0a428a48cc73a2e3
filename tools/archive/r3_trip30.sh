#!/bin/bash
# views with runs of 32 / 96 / 160 + tail split on two-level rows: tests, D = 96 rows
set -u
O=gpurun_out/${1:-r3t30}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 600 -k "gemm or tensordot" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_gemm.log
timeout 600 python - <<'PY' | tee $O/d96_views.txt
import time, numpy as np, tensornetwork_amd as ta, bench
from tensornetwork_amd import _lib
be = ta.get_hip_backend(); lib = be.lib
ta.configure_gc(freeze=True)
for D in (96, 160):
  A, B = bench.make_nodes(ta, be, D, "L0", seed=7, fill="normal")
  for rep in range(2):
    for layout in ("L0", "L1"):
      t, perm = bench.timed_steps(be, lambda: bench.one_step(ta, be, A, B, layout), 10, batches=3)
      print("D=%d %s  %.3f ms %.0f TF permutes %.1f %s" % (D, layout, t*1e3, 2.0*D**6/t/1e12, perm, lib.tnh_gemm_last_kernel().decode()), flush=True)
  del A, B
PY
