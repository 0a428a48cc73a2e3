"""f32 GEMM on the bf16 matrix cores (3 x bf16 split, six products) vs the native f32 MFMA kernel:
error against float64 and time.   python tools/f32_split_probe.py [--perf]     (GPU box only)"""
import argparse, ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor

ap = argparse.ArgumentParser()
ap.add_argument("--perf", action="store_true")
a = ap.parse_args()
be = ta.get_hip_backend()


def gemm(variant, A, B, m, n, k, ta_, tb_, iters=0):
  c = DeviceTensor.empty((m, n), _lib.F32)
  lda = m if ta_ else k
  ldb = k if tb_ else n
  _lib.check(be.lib.tnh_gemm_set_variant(variant.encode()))
  def call():
    _lib.check(be.lib.tnh_gemm(_lib.F32, _lib.F32, ta_, tb_, m, n, k, ctypes.c_void_p(A.ptr), lda,
                               ctypes.c_void_p(B.ptr), ldb, ctypes.c_void_p(c.ptr), n, 1, 0, 0, 0))
  try:
    call()
    name = be.lib.tnh_gemm_last_kernel().decode()
    ms = None
    if iters:
      call()
      s = _lib.Event().record()
      for _ in range(iters):
        call()
      e = _lib.Event().record()
      e.synchronize()
      ms = s.elapsed_ms(e) / iters
  finally:
    _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
  return c, name, ms


rng = np.random.default_rng(0)
m, n, k = 4096, 4096, 1096          # K not a multiple of 64: exercises the zero fill
for ta_, tb_ in ((0, 1), (0, 0), (1, 0), (1, 1)):
  ah = (rng.standard_normal((k, m) if ta_ else (m, k)) * np.exp(rng.uniform(-3, 3, (k, m) if ta_ else (m, k)))).astype(np.float32)
  bh = rng.standard_normal((n, k) if tb_ else (k, n)).astype(np.float32)
  A, B = be.convert_to_tensor(ah), be.convert_to_tensor(bh)
  a64 = (ah.T if ta_ else ah).astype(np.float64)
  b64 = (bh.T if tb_ else bh).astype(np.float64)
  exact = a64 @ b64
  den = np.abs(a64) @ np.abs(b64)
  out = {}
  for variant in ("auto", "auto:s0"):
    c, name, _ = gemm(variant, A, B, m, n, k, ta_, tb_)
    out[name] = float((np.abs(np.asarray(c).astype(np.float64) - exact) / den).max())
  print(json.dumps({"layout": [ta_, tb_], "m": m, "n": n, "k": k, "max_err_over_sum_abs": out}), flush=True)

if a.perf:
  for (m, n, k, iters) in [(4096, 4096, 4096, 10), (8192, 8192, 8192, 5), (16384, 16384, 4096, 3)]:
    for ta_, tb_ in ((0, 1), (0, 0)):
      A = be.device_random((m * k,), dtype=np.float32, seed=1, normal=True, b=1.0)
      B = be.device_random((n * k,), dtype=np.float32, seed=2, normal=True, b=1.0)
      for variant in ("auto", "auto:s0"):
        _, name, ms = gemm(variant, A, B, m, n, k, ta_, tb_, iters)
        print(json.dumps({"m": m, "n": n, "k": k, "layout": [ta_, tb_], "kernel": name, "ms": ms,
                          "f32_tflops": 2.0 * m * n * k / ms / 1e9}), flush=True)
