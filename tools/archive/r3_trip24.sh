#!/bin/bash
# persistent ping-pong GEMM: tests, then A/B over shapes
set -u
O=gpurun_out/${1:-r3t24}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 600 -k "gemm or tensordot" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_gemm.log
timeout 600 python - <<'PY' | tee $O/persist_ab.txt
import ctypes, numpy as np, tensornetwork_amd as ta
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
shapes = [(4096,4096,4096,20),(9216,9216,9216,5),(16384,16384,16384,3),(8192,8192,65536,3),(65536,32768,1024,3),(1048576,2048,1024,3),(36864,36864,4096,2),(65536,65536,8192,1)]
for (m,n,k,it) in shapes:
  A = be.device_random((m,k), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=k**-0.5)
  B = be.device_random((n,k), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=k**-0.5)
  C = DeviceTensor.empty((m,n), _lib.BF16)
  res = {}
  for v in ("auto:g0","auto:g1","auto:g0","auto:g1"):
    _lib.check(lib.tnh_gemm_set_variant(v.encode()))
    ms = timed(lambda: _lib.check(lib.tnh_gemm(_lib.BF16,_lib.BF16,0,1,m,n,k,vp(A),k,vp(B),k,vp(C),n,1,0,0,0)), it)
    res.setdefault(v, []).append(2.0*m*n*k/ms/1e9)
  _lib.check(lib.tnh_gemm_set_variant(b"auto"))
  print("%8d x %8d x %8d  per-tile %s TF   persistent %s TF   %s" % (m,n,k, ["%.0f"%x for x in res["auto:g0"]], ["%.0f"%x for x in res["auto:g1"]], lib.tnh_gemm_last_kernel().decode()), flush=True)
  del A,B,C
PY
