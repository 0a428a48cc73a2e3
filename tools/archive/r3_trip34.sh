#!/bin/bash
# block-cyclic row reduction + size-aware streaming grids: tests and helper rows
set -u
O=gpurun_out/${1:-r3t34}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 600 -k "reduc or sum or norm or trace or permute or transpose or elementwise or unary or binary or broadcast" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log
for i in 1 2; do
timeout 300 python - <<'PY' | tee -a $O/helpers.txt
import bench, tensornetwork_amd as ta, numpy as np
be = ta.get_hip_backend()
x = be.device_random((128,)*4, dtype=ta.bfloat16, seed=42)
for _ in range(60): y = be.transpose(x, (0,2,1,3))
del x, y
for r in bench.helpers_bench(ta, be):
  print("%-78s %8.3f ms %7.0f GB/s" % (r["op"][:78], r["ms"], r["gbps"]))
x = be.device_random((1 << 28,), dtype=np.float32, seed=3)
import time
from tensornetwork_amd import _lib
for name, fn in (("norm 1 GiB", lambda: be.norm(x)), ("sum 1 GiB", lambda: be.sum(x))):
  fn(); s = _lib.Event().record()
  for _ in range(10): fn()
  e = _lib.Event().record(); e.synchronize()
  print("%-78s %8.3f ms %7.0f GB/s" % (name, s.elapsed_ms(e)/10, x.nbytes / (s.elapsed_ms(e)/10) / 1e6))
PY
done
