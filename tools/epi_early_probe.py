"""A/B of the ping-pong GEMM's early start of the next tile under the draining epilogue stores (knob ":e0" = off).
  python tools/epi_early_probe.py [--iters 6]
Parity first (bit-identical: only a wait changes), then kernel-only TFLOP/s per shape, arms alternating; JSON lines."""
import argparse, ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
be = ta.get_hip_backend()


def call(view, out_dt, m, n, k, A, B, C):
  if view:
    va, vb = _lib.OperandView(m, k, 0, k, 1, 0), _lib.OperandView(n, k, 0, k, 1, 0)
    _lib.check(be.lib.tnh_gemm_view(_lib.BF16, out_dt, m, n, k, ctypes.c_void_p(A.ptr), ctypes.byref(va),
                                    ctypes.c_void_p(B.ptr), ctypes.byref(vb), ctypes.c_void_p(C.ptr), n))
  else:
    _lib.check(be.lib.tnh_gemm(_lib.BF16, out_dt, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), k, ctypes.c_void_p(B.ptr), k,
                               ctypes.c_void_p(C.ptr), n, 1, 0, 0, 0))


def run(knob, view, out_dt, m, n, k, A, B):
  C = DeviceTensor.empty((m, n), out_dt)
  _lib.check(be.lib.tnh_gemm_set_variant(knob.encode()))
  try:
    call(view, out_dt, m, n, k, A, B, C)
    be.synchronize()
  finally:
    _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
  return np.asarray(C).copy()


for (m, n, k) in [(8192, 8192, 128), (8192, 8192, 512), (8192 + 40, 8192 - 24, 1088), (16384, 8192, 192), (12288, 12288, 4096)]:
  A = be.device_random((m, k), dtype=ta.bfloat16, seed=11, normal=True, a=0.0, b=1.0)
  B = be.device_random((n, k), dtype=ta.bfloat16, seed=12, normal=True, a=0.0, b=1.0)
  for view in (False, True):
    for out_dt in (_lib.BF16, _lib.F32):
      ref = run("auto:e0", view, out_dt, m, n, k, A, B)
      for rep in range(3):
        got = run("auto", view, out_dt, m, n, k, A, B)
        print(json.dumps({"parity": [m, n, k], "view": view, "out_f32": out_dt == _lib.F32, "rep": rep,
                          "bit_identical": bool(np.array_equal(got, ref))}), flush=True)
  del A, B

for (m, n, k) in [(32768, 32768, 512), (32768, 32768, 1024), (32768, 32768, 2048), (32768, 32768, 4096), (1048576, 32768, 1024),
                  (20736, 20736, 1728), (65536, 65536, 4096), (262144, 4096, 4096), (16384, 16384, 16384)]:
  A = be.device_random((m, k), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=float(k) ** -0.5)
  B = be.device_random((n, k), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=1.0)
  C = DeviceTensor.empty((m, n), _lib.BF16)
  for rnd in range(a.rounds):
    for knob in ("auto:e0", "auto"):
      _lib.check(be.lib.tnh_gemm_set_variant(knob.encode()))
      call(True, _lib.BF16, m, n, k, A, B, C)
      be.synchronize()
      s = _lib.Event().record()
      for _ in range(a.iters):
        call(True, _lib.BF16, m, n, k, A, B, C)
      e = _lib.Event().record()
      e.synchronize()
      _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
      ms = s.elapsed_ms(e) / a.iters
      print(json.dumps({"gemm": [m, n, k], "knob": knob, "round": rnd, "ms": round(ms, 4),
                        "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}), flush=True)
  del A, B, C
