#!/bin/bash
# staggered XCD start A/B on short-K persistent launches
set -u
O=gpurun_out/${1:-r3t26}
mkdir -p $O
for st in 0 1 0 1; do
TNH_GEMM_STAGGER=$st timeout 600 python - <<'PY' | tee -a $O/stagger.txt
import ctypes, os, numpy as np, tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor
be = ta.get_hip_backend(); lib = be.lib
vp = lambda t: ctypes.c_void_p(t.ptr)
def timed(fn, iters):
  for _ in range(2): fn()
  s = _lib.Event().record()
  for _ in range(iters): fn()
  e = _lib.Event().record(); e.synchronize()
  return s.elapsed_ms(e) / iters
m, n = 32768, 32768
out = []
for k in (256, 1024, 2048, 4096):
    sc = k**-0.5
    A = be.device_random((m,k), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=sc)
    B = be.device_random((n,k), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=sc)
    C = DeviceTensor.empty((m,n), _lib.BF16)
    ms = timed(lambda: _lib.check(lib.tnh_gemm(_lib.BF16,_lib.BF16,0,1,m,n,k,vp(A),k,vp(B),k,vp(C),n,1,0,0,0)), max(2, 8192 // k))
    out.append("K=%d %.0f TF %.2f us/tile" % (k, 2.0*m*n*k/ms/1e9, ms * 1e3 / 64))
    del A, B, C
print("stagger=%s  %s" % (os.environ["TNH_GEMM_STAGGER"], "   ".join(out)), flush=True)
PY
done
