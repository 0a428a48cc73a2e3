#!/bin/bash
# 16-byte epilogue stores: GEMM tests, then tile time vs K (persistent / per-tile)
set -u
O=gpurun_out/${1:-r3t25}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 600 -k "gemm or tensordot" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_gemm.log
timeout 600 python - <<'PY' | tee $O/shortk2.txt
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
m, n = 32768, 32768
for k in (256, 512, 1024, 2048, 4096, 16384):
    sc = k**-0.5
    A = be.device_random((m,k), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=sc)
    B = be.device_random((n,k), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=sc)
    C = DeviceTensor.empty((m,n), _lib.BF16)
    out = []
    for v in ("auto:g0","auto:g1"):
      _lib.check(lib.tnh_gemm_set_variant(v.encode()))
      ms = timed(lambda: _lib.check(lib.tnh_gemm(_lib.BF16,_lib.BF16,0,1,m,n,k,vp(A),k,vp(B),k,vp(C),n,1,0,0,0)), max(2, 8192 // k))
      tiles = (m // 256) * (n // 256)
      out.append("%s %.0f TF, %.2f us per tile-wave" % (v, 2.0*m*n*k/ms/1e9, ms * 1e3 / (tiles / 256)))
    _lib.check(lib.tnh_gemm_set_variant(b"auto"))
    print("K=%6d  %s" % (k, "   ".join(out)), flush=True)
    del A, B, C
_lib.check(lib.tnh_gemm_set_variant(b"bf16_256pp"))
for (m, n) in ((1024, 1024), (4096, 4096)):
  for k in (1024, 4096):
    sc = k**-0.5
    A = be.device_random((m,k), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=sc)
    B = be.device_random((n,k), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=sc)
    C = DeviceTensor.empty((m,n), _lib.BF16)
    for _ in range(20): lib.tnh_gemm(_lib.BF16,_lib.BF16,0,1,m,n,k,vp(A),k,vp(B),k,vp(C),n,1,0,0,0)
    ms = timed(lambda: _lib.check(lib.tnh_gemm(_lib.BF16,_lib.BF16,0,1,m,n,k,vp(A),k,vp(B),k,vp(C),n,1,0,0,0)), 200)
    print("tiles=%4d K=%5d  %.2f us per launch" % ((m // 256) * (n // 256), k, ms * 1e3), flush=True)
PY
