#!/bin/bash
# per-XCD K-walk phase on the bond-sweep cubes
set -u
O=gpurun_out/${1:-r3t26}
mkdir -p $O
rm -f $O/kphase_cubes.txt
for ph in 0 1 0 1; do
TNH_GEMM_KPHASE=$ph timeout 900 python - <<'PY' | tee -a $O/kphase_cubes.txt
import ctypes, os, numpy as np, tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor
be = ta.get_hip_backend(); lib = be.lib
vp = lambda t: ctypes.c_void_p(t.ptr)
def timed(fn, iters):
  for _ in range(1): fn()
  s = _lib.Event().record()
  for _ in range(iters): fn()
  e = _lib.Event().record(); e.synchronize()
  return s.elapsed_ms(e) / iters
out = []
for (d, it) in ((4096, 50), (9216, 10), (16384, 5), (36864, 2), (65536, 2)):
      m = n = k = d
      sc = k**-0.5
      A = be.device_random((m, k), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=sc)
      B = be.device_random((n, k), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=sc)
      C = DeviceTensor.empty((m,n), _lib.BF16)
      ms = timed(lambda: _lib.check(lib.tnh_gemm(_lib.BF16,_lib.BF16,0,1,m,n,k,vp(A),k,vp(B),k,vp(C),n,1,0,0,0)), it)
      out.append("%d^3 %.0f" % (d, 2.0*m*n*k/ms/1e9))
      del A, B, C
print("kphase=%s TF: %s" % (os.environ["TNH_GEMM_KPHASE"], "   ".join(out)), flush=True)
PY
done
