"""D = 96 shape (9216^3, 36 x 36 tiles): view GEMM with the tail split-K on / off, alternating in one process.
  python tools/tail_probe.py"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor
be = ta.get_hip_backend()
for (m, n, k) in [(9216, 9216, 9216), (9216, 9216, 2048), (7680, 7680, 8192)]:
  a = be.device_random((m, k), dtype=ta.bfloat16, seed=1, normal=True, b=k ** -0.5)
  b = be.device_random((n, k), dtype=ta.bfloat16, seed=2, normal=True, b=1.0)
  c = DeviceTensor.empty((m, n), _lib.BF16)
  va = _lib.OperandView(m, k, 0, k, 1, 0)
  vb = _lib.OperandView(n, k, 0, k, 1, 0)
  def call():
    _lib.check(be.lib.tnh_gemm_view(_lib.BF16, _lib.BF16, m, n, k, ctypes.c_void_p(a.ptr), ctypes.byref(va), ctypes.c_void_p(b.ptr),
                                    ctypes.byref(vb), ctypes.c_void_p(c.ptr), n))
  for rep in range(3):
    for knob in (b"auto", b"auto:t0"):
      _lib.check(be.lib.tnh_gemm_set_variant(knob))
      call(); call()
      s = _lib.Event().record()
      for _ in range(10):
        call()
      e = _lib.Event().record(); e.synchronize()
      ms = s.elapsed_ms(e) / 10
      print(json.dumps({"m": m, "n": n, "k": k, "knob": knob.decode(), "rep": rep, "ms": ms, "tflops": 2.0 * m * n * k / ms / 1e9,
                        "kernel": be.lib.tnh_gemm_last_kernel().decode()}), flush=True)
  _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
