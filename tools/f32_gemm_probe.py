"""Native f32 / f64 GEMM rates by layout and size, old vs new f32 kernel (GPU box only).
  python tools/f32_gemm_probe.py"""
import ctypes, json, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor

be = ta.get_hip_backend()


def rate(code, ta_, tb_, m, n, k, iters=5):
  dt = np.float32 if code == _lib.F32 else np.float64
  a = be.device_random((m * k,), dtype=dt, seed=1, normal=True)
  b = be.device_random((n * k,), dtype=dt, seed=2, normal=True)
  c = DeviceTensor.empty((m, n), code)
  lda = m if ta_ else k
  ldb = k if tb_ else n
  _lib.check(be.lib.tnh_gemm_set_variant(b"auto:s0"))     # keep f32 on the f32 matrix instruction
  def call():
    _lib.check(be.lib.tnh_gemm(code, code, ta_, tb_, m, n, k, ctypes.c_void_p(a.ptr), lda, ctypes.c_void_p(b.ptr), ldb,
                               ctypes.c_void_p(c.ptr), n, 1, 0, 0, 0))
  call(); call()
  s = _lib.Event().record()
  for _ in range(iters):
    call()
  e = _lib.Event().record(); e.synchronize()
  ms = s.elapsed_ms(e) / iters
  name = be.lib.tnh_gemm_last_kernel().decode()
  _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
  return {"dtype": "f32" if code == _lib.F32 else "f64", "tA": ta_, "tB": tb_, "m": m, "n": n, "k": k, "ms": ms,
          "tflops": 2.0 * m * n * k / ms / 1e9, "kernel": name}


for (m, n, k) in [(4096, 4096, 4096), (2048, 2048, 2048), (8192, 8192, 1024), (1024, 512, 1024), (65536, 256, 32)]:
  for ta_, tb_ in [(0, 1), (0, 0), (1, 0), (1, 1)]:
    print(json.dumps(rate(_lib.F32, ta_, tb_, m, n, k)), flush=True)
for (m, n, k) in [(4096, 4096, 4096), (2048, 2048, 2048)]:
  for ta_, tb_ in [(0, 1), (0, 0)]:
    print(json.dumps(rate(_lib.F64, ta_, tb_, m, n, k)), flush=True)
