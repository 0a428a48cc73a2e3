#!/bin/bash
# line-mate gather: tests + A/B on the f32 rank-6 permute
set -u
O=gpurun_out/${1:-r3t10}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 600 -k "permute or transpose" > $O/pytest_layout.log 2>&1; echo "pytest rc=$?" | tee $O/trip.log
tail -3 $O/pytest_layout.log
rm -f $O/gather_ab.txt
for lm in 0 1 0 1; do
TNH_PERMUTE_LINEMATE=$lm timeout 300 python - <<PY | tee -a $O/gather_ab.txt
import numpy as np, tensornetwork_amd as ta
from tensornetwork_amd import hip_backend, _lib
be = hip_backend.HipBackend()
x = be.device_random((16,) * 6, dtype=np.float32, seed=41)
y = be.device_random((32, 32, 32, 32, 8), dtype=np.float32, seed=42)
z = be.device_random((16,) * 6, dtype=ta.bfloat16, seed=43)
def t(fn, nbytes, reps=200):
  for _ in range(300): fn()
  be.synchronize()
  s = _lib.Event().record()
  for _ in range(reps): fn()
  e = _lib.Event().record(); e.synchronize()
  ms = s.elapsed_ms(e) / reps
  return nbytes / ms / 1e6
print("linemate=$lm f32 16^6 (0,2,4,1,3,5): %.0f GB/s   f32 32^4x8 (0,2,1,3,4): %.0f GB/s   bf16 16^6 (0,2,4,1,3,5): %.0f GB/s" % (
  t(lambda: be.transpose(x, (0, 2, 4, 1, 3, 5)), 2 * x.nbytes), t(lambda: be.transpose(y, (0, 2, 1, 3, 4)), 2 * y.nbytes),
  t(lambda: be.transpose(z, (0, 2, 4, 1, 3, 5)), 2 * z.nbytes)))
PY
done
