"""In-place (view) GEMM vs permute + NT vs padded NT on the BASELINE shapes (GPU box only).

  python tools/view_probe.py [--shapes 65536x65536x65536,8192x8192x262144] [--iters 3]

Arms per shape (random N(0, 1/sqrt(K)) bf16 operands, a = [M][K], b = [K][N] as config-2 L0 stores them):
  nt_copy      what round 1 did: K1 permute of b to [N][K], then the NT kernel           (permute timed too)
  nn_view      tnh_gemm_view: b read in place as a k-major operand (LDS transpose reads)  (no permute)
  nt_pad_b     permute of b into rows padded by 64 elements (power-of-two row strides alias HBM channels)
  nt_pad_ab    both operands in padded copies (a copied once: +1 pass over a, timed)
Prints one JSON line per arm: gemm-only ms / TFLOP/s and whole-path ms / TFLOP/s."""
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
ap.add_argument("--shapes", default="8192x8192x65536,8192x8192x262144,65536x65536x65536")
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--pad", type=int, default=64)
a = ap.parse_args()
be = ta.get_hip_backend()
lib = be.lib
vp = lambda t: ctypes.c_void_p(t.ptr)


def timed(fn, iters):
  fn()
  s = _lib.Event().record()
  for _ in range(iters):
    fn()
  e = _lib.Event().record()
  e.synchronize()
  return s.elapsed_ms(e) / iters


def padded_copy(dst, src, rows, cols, ld):
  """dst[r, :cols] = src[r, :] with row stride ld (one strided pass)."""
  _lib.check(lib.tnh_strided_scatter(vp(dst), vp(src), 2, _lib.i64_array((rows, cols)), _lib.i64_array((ld, 1)), 0, 2),
             "tnh_strided_scatter")


for shape in a.shapes.split(","):
  m, n, k = (int(x) for x in shape.split("x"))
  sc = float(k) ** -0.5
  A = be.device_random((m, k), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=sc)
  B = be.device_random((k, n), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=sc)
  C = DeviceTensor.empty((m, n), _lib.BF16)
  flops = 2.0 * m * n * k
  ld = k + a.pad

  def report(arm, gemm_ms, path_ms, kernel):
    print(json.dumps({"m": m, "n": n, "k": k, "arm": arm, "gemm_ms": gemm_ms, "gemm_tflops": flops / gemm_ms / 1e9,
                      "path_ms": path_ms, "path_tflops": flops / path_ms / 1e9, "kernel": kernel}), flush=True)

  # ---- nt_copy
  Bt = be.transpose(B, (1, 0))
  gemm = lambda: _lib.check(lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, vp(A), k, vp(Bt), k, vp(C), n, 1, 0, 0, 0))
  g = timed(gemm, a.iters)
  kern = lib.tnh_gemm_last_kernel().decode()
  def path():
    bt = be.transpose(B, (1, 0))
    _lib.check(lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, vp(A), k, vp(bt), k, vp(C), n, 1, 0, 0, 0))
  report("nt_copy", g, timed(path, a.iters), kern)
  ref = np.asarray(be.getitem(C, (slice(0, 4), slice(0, 8))))
  # ---- nn_view
  va = _lib.OperandView(m, k, 0, k, 1, 0)
  vb = _lib.OperandView(n, 1, 0, k, n, 0)
  view = lambda: _lib.check(lib.tnh_gemm_view(_lib.BF16, _lib.BF16, m, n, k, vp(A), ctypes.byref(va), vp(B), ctypes.byref(vb),
                                              vp(C), n))
  g = timed(view, a.iters)
  same = bool(np.array_equal(np.asarray(be.getitem(C, (slice(0, 4), slice(0, 8)))), ref))
  report("nn_view" + ("" if same else "_MISMATCH"), g, g, lib.tnh_gemm_last_kernel().decode())
  # ---- nt_pad_b
  Btp = DeviceTensor.empty((n, ld), _lib.BF16)
  padded_copy(Btp, Bt, n, k, ld)
  gemm = lambda: _lib.check(lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, vp(A), k, vp(Btp), ld, vp(C), n, 1, 0, 0, 0))
  g = timed(gemm, a.iters)
  def path():
    bt = be.transpose(B, (1, 0))
    padded_copy(Btp, bt, n, k, ld)
    _lib.check(lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, vp(A), k, vp(Btp), ld, vp(C), n, 1, 0, 0, 0))
  report("nt_pad_b", g, timed(path, a.iters), lib.tnh_gemm_last_kernel().decode())
  del Bt
  # ---- nt_pad_ab
  Ap = DeviceTensor.empty((m, ld), _lib.BF16)
  padded_copy(Ap, A, m, k, ld)
  gemm = lambda: _lib.check(lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, vp(Ap), ld, vp(Btp), ld, vp(C), n, 1, 0, 0, 0))
  g = timed(gemm, a.iters)
  def path():
    bt = be.transpose(B, (1, 0))
    padded_copy(Btp, bt, n, k, ld)
    padded_copy(Ap, A, m, k, ld)
    _lib.check(lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, vp(Ap), ld, vp(Btp), ld, vp(C), n, 1, 0, 0, 0))
  report("nt_pad_ab", g, timed(path, a.iters), lib.tnh_gemm_last_kernel().decode())
  same = bool(np.array_equal(np.asarray(be.getitem(C, (slice(0, 4), slice(0, 8)))), ref))
  if not same:
    print(json.dumps({"warning": "padded result differs from nt_copy"}), flush=True)
  del A, B, C, Btp, Ap
  _lib.check(lib.tnh_trim())
