"""k-major operand read in place (tnh_gemm_view) against the NT kernel on a permuted copy; GPU box only.
  python tools/kmajor_touch_probe.py [--shapes 8192x8192x262144] [--iters 5] [--zeros] [--pitch-mult 4] [--variant auto:r1]
(Round 3 also ran it with a look-ahead page touch in the kernel -- TNH_KMAJOR_TOUCH=D, one wave reading 4 B of each
k-row of K-tile t + D -- which cost 2 %; the code was removed, the numbers are in profiles/r03_kmajor_inplace.md.)"""
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

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="8192x8192x262144")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--pitch-mult", type=int, default=1, help="b is the first n columns of a [k][n * mult] matrix (same bytes read, mult x the span)")
ap.add_argument("--variant", default="auto", help="tnh_gemm_set_variant string, e.g. auto:r1")
ap.add_argument("--zeros", action="store_true", help="all-zero operands: no data-dependent power, the clock stays up")
a = ap.parse_args()
be = ta.get_hip_backend()
lib = be.lib
_lib.check(lib.tnh_gemm_set_variant(a.variant.encode()), 'tnh_gemm_set_variant')
vp = lambda t: ctypes.c_void_p(t.ptr)


def timed(fn, iters):
  for _ in range(2):
    fn()
  s = _lib.Event().record()
  for _ in range(iters):
    fn()
  e = _lib.Event().record()
  e.synchronize()
  return s.elapsed_ms(e) / iters


for shape in a.shapes.split(","):
  m, n, k = (int(x) for x in shape.split("x"))
  sc = 0.0 if a.zeros else float(k) ** -0.5
  A = be.device_random((m, k), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=sc)
  B = be.device_random((k, n), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=sc)
  C = DeviceTensor.empty((m, n), _lib.BF16)
  C2 = DeviceTensor.empty((m, n), _lib.BF16)
  flops = 2.0 * m * n * k
  Bt = be.transpose(B, (1, 0))
  gemm = lambda: _lib.check(lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, vp(A), k, vp(Bt), k, vp(C), n, 1, 0, 0, 0))
  g_nt = timed(gemm, a.iters)
  va = _lib.OperandView(m, k, 0, k, 1, 0)
  pm = a.pitch_mult
  Bv = B
  if pm > 1:
    Bv = DeviceTensor.empty((k, n * pm), _lib.BF16)
    _lib.check(lib.tnh_strided_scatter(vp(Bv), vp(B), 2, _lib.i64_array((k, n)), _lib.i64_array((n * pm, 1)), 0, 2),
               "tnh_strided_scatter")
  vb = _lib.OperandView(n, 1, 0, k, n * pm, 0)
  view = lambda: _lib.check(lib.tnh_gemm_view(_lib.BF16, _lib.BF16, m, n, k, vp(A), ctypes.byref(va), vp(Bv), ctypes.byref(vb),
                                              vp(C2), n))
  g_view = timed(view, a.iters)
  same = bool(np.array_equal(np.asarray(C), np.asarray(C2)))
  perm = timed(lambda: be.transpose(B, (1, 0)), a.iters)
  print(json.dumps({"touch": int(os.environ.get("TNH_KMAJOR_TOUCH", "0")), "zeros": a.zeros, "variant": a.variant, "pitch_mult": a.pitch_mult, "m": m, "n": n, "k": k, "nt_ms": g_nt,
                    "nt_tflops": flops / g_nt / 1e9, "permute_ms": perm, "nt_path_tflops": flops / (g_nt + perm) / 1e9,
                    "view_ms": g_view, "view_tflops": flops / g_view / 1e9, "bit_identical": same,
                    "kernel": lib.tnh_gemm_last_kernel().decode()}), flush=True)
