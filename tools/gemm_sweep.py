"""Kernel-level sweep of the K2/K1 kernels with HIP-event timing (GPU box only).

  python tools/gemm_sweep.py [--quick]

Prints one JSON line per (kernel variant, shape, fill): TFLOP/s or GB/s."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta  # noqa: E402
from tensornetwork_amd import _lib  # noqa: E402
from tensornetwork_amd.device_tensor import DeviceTensor  # noqa: E402


def time_calls(fn, iters, warm=2):
  for _ in range(warm):
    fn()
  s = _lib.Event().record()
  for _ in range(iters):
    fn()
  e = _lib.Event().record()
  e.synchronize()
  return s.elapsed_ms(e) / iters


def gemm_tflops(be, code, out_code, ta_, tb_, m, n, k, variant, fill, iters):
  if fill == "zeros":
    a = be.zeros((m * k,), dtype=ta.bfloat16 if code == _lib.BF16 else np.float32)
    b = be.zeros((n * k,), dtype=ta.bfloat16 if code == _lib.BF16 else np.float32)
  else:
    dt = {_lib.BF16: ta.bfloat16, _lib.F16: np.float16, _lib.F32: np.float32, _lib.F64: np.float64}[code]
    a = be.device_random((m * k,), dtype=dt, seed=1, normal=False, a=-1.0, b=1.0)
    b = be.device_random((n * k,), dtype=dt, seed=2, normal=False, a=-1.0, b=1.0)
  c = DeviceTensor.empty((m, n), out_code)
  lda = m if ta_ else k
  ldb = k if tb_ else n
  _lib.check(be.lib.tnh_gemm_set_variant(variant.encode()))

  def call():
    _lib.check(be.lib.tnh_gemm(code, out_code, ta_, tb_, m, n, k, ctypes.c_void_p(a.ptr), lda,
                               ctypes.c_void_p(b.ptr), ldb, ctypes.c_void_p(c.ptr), n, 1, 0, 0, 0))
  try:
    ms = time_calls(call, iters)
    name = be.lib.tnh_gemm_last_kernel().decode()
  finally:
    _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
  return {"op": "gemm", "kernel": name, "variant": variant, "m": m, "n": n, "k": k, "fill": fill, "ms": ms,
          "tflops": 2.0 * m * n * k / ms / 1e9}


def permute_gbps(be, shape, perm, dtype, iters):
  x = be.device_random(shape, dtype=dtype, seed=5)
  def call():
    be.transpose(x, perm)
  ms = time_calls(call, iters)
  return {"op": "permute", "shape": list(shape), "perm": list(perm), "dtype": str(dtype), "ms": ms,
          "gbps": 2.0 * x.nbytes / ms / 1e6}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--quick", action="store_true")
  ap.add_argument("--gemm-only", action="store_true")
  ap.add_argument("--skinny", action="store_true",
                  help="ragged / skinny NT shapes of the D=12 sliced network: generic vs ragged matrix-core kernel")
  ap.add_argument("--stream", action="store_true",
                  help="small x very long NT products: streaming kernel (auto) vs the ragged tile kernels")
  ap.add_argument("--variants", default="bf16_128,bf16_256,bf16_256pp")
  args = ap.parse_args()
  global VARIANTS
  VARIANTS = args.variants.split(',')
  be = ta.get_hip_backend()
  if args.stream:
    shapes = [(144, 2985984, 144), (2985984, 144, 144), (144, 248832, 144), (248832, 144, 144), (64, 4000000, 64),
              (192, 2000000, 192), (2000000, 192, 192), (128, 2985984, 128), (2985984, 128, 128), (80, 3000000, 80),
              (96, 3000000, 96), (112, 3000000, 112), (3000000, 96, 96), (160, 2000000, 160), (176, 2000000, 64),
              (144, 2985984, 16), (144, 2985984, 64), (144, 2985984, 192), (169, 4826816, 168)]
    for (m, n, k) in shapes:
      for variant in ("bf16_ragged", "auto"):
        rec = gemm_tflops(be, _lib.BF16, _lib.BF16, 0, 1, m, n, k, variant, "uniform", 5)
        rec["gbps"] = 2.0 * (m * k + n * k + m * n) / rec["ms"] / 1e6
        print(json.dumps(rec), flush=True)
    return
  if args.skinny:
    shapes = [(144, 2985984, 144), (2985984, 144, 144), (144, 248832, 1728), (248832, 144, 1728),
              (1728, 248832, 12), (248832, 1728, 12), (144, 248832, 144), (248832, 144, 144),
              (4000, 4000, 4000), (12, 20736, 1728), (20736, 12, 1728)]
    for (m, n, k) in shapes:
      for variant in ("bf16_ragged_128x128", "bf16_ragged_192x128", "bf16_ragged_128x192", "bf16_ragged_256x64", "auto"):
        if (variant.endswith("192x128") and not 128 < m <= 192) or (variant.endswith("128x192") and not 128 < n <= 192) or \
           (variant.endswith("256x64") and not m <= 256 < n):
          continue
        if variant == "generic" and 2.0 * m * n * k > 3e11:
          continue
        rec = gemm_tflops(be, _lib.BF16, _lib.BF16, 0, 1, m, n, k, variant, "uniform", 5)
        rec["gbps"] = 2.0 * (m * k + n * k + m * n) / rec["ms"] / 1e6
        print(json.dumps(rec), flush=True)
    return
  sizes = [4096] if args.quick else [2048, 4096, 8192]
  for n in sizes:
    for variant in VARIANTS:
      for fill in ("uniform", "zeros"):
        print(json.dumps(gemm_tflops(be, _lib.BF16, _lib.BF16, 0, 1, n, n, n, variant, fill, 10)), flush=True)
  # north-star D=512 row: M=N=8192, K=262144
  if not args.quick:
    for variant in VARIANTS:
      print(json.dumps(gemm_tflops(be, _lib.BF16, _lib.BF16, 0, 1, 8192, 8192, 262144, variant, "uniform", 3)), flush=True)
  if args.gemm_only:
    return
  for n in ([2048] if args.quick else [2048, 4096]):
    print(json.dumps(gemm_tflops(be, _lib.F32, _lib.F32, 0, 0, n, n, n, "generic", "uniform", 5)), flush=True)
    print(json.dumps(gemm_tflops(be, _lib.F32, _lib.F32, 0, 1, n, n, n, "generic", "uniform", 5)), flush=True)
    print(json.dumps(gemm_tflops(be, _lib.F64, _lib.F64, 0, 0, n, n, n, "generic", "uniform", 5)), flush=True)
    print(json.dumps(gemm_tflops(be, _lib.BF16, _lib.BF16, 0, 0, n, n, n, "generic", "uniform", 5)), flush=True)
  for shape, perm, dt in [((8192, 8192), (1, 0), np.float32), ((16384, 16384), (1, 0), ta.bfloat16),
                          ((256, 256, 256, 16), (2, 3, 0, 1), ta.bfloat16), ((16,) * 6, (0, 2, 4, 1, 3, 5), np.float32),
                          ((64, 64, 64, 64), (0, 2, 1, 3), np.float32), ((512, 2, 512, 64), (2, 1, 0, 3), np.float32),
                          ((4096, 4096, 2), (1, 0, 2), np.float32)]:
    print(json.dumps(permute_gbps(be, shape, perm, dt, 10)), flush=True)


if __name__ == "__main__":
  main()
