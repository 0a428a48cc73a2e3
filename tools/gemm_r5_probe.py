"""Round-5 A/B of bf16 GEMM schedule variants on one box (GPU only).

  python tools/gemm_r5_probe.py [--seconds 1.5] [--variants auto,bf16_256pp:p7,...] [--headline 1]

Step 1 (parity): every variant against the default kernel on a ragged-free 1024 x 768 x 4096 and a persistent
2560 x 2560 x 1024 product -- bit-identical for the 16x16x32 schedules, within bf16 rounding for the 32x32x16 ones
(different summation order).  Step 2 (power): tools/power_probe.run_arm per (shape, variant, fill), random and zero
operands, with board power and shader clock.  Step 3 (--headline): the headline 65536^3 view GEMM per variant that
the view kernel has (auto, :p7).  One JSON line per arm."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tensornetwork_amd as ta  # noqa: E402
from tensornetwork_amd import _lib  # noqa: E402
from tensornetwork_amd.device_tensor import DeviceTensor  # noqa: E402
from tensornetwork_amd.telemetry import Sampler, Telemetry  # noqa: E402
import power_probe  # noqa: E402


def parity(be, variants):
  out = []
  for (m, n, k) in [(1024, 768, 4096), (2560, 2560, 1024), (8192, 8192, 512)]:
    A = be.device_random((m, k), dtype=ta.bfloat16, seed=11, normal=True, a=0.0, b=1.0)
    B = be.device_random((n, k), dtype=ta.bfloat16, seed=12, normal=True, a=0.0, b=1.0)
    res = {}
    for v in variants:
      C = DeviceTensor.empty((m, n), _lib.BF16)
      _lib.check(be.lib.tnh_gemm_set_variant(v.encode()))
      try:
        _lib.check(be.lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), k, ctypes.c_void_p(B.ptr), k,
                                   ctypes.c_void_p(C.ptr), n, 1, 0, 0, 0))
        be.synchronize()
        kern = be.lib.tnh_gemm_last_kernel().decode()
      finally:
        _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
      res[v] = (np.asarray(C).copy(), kern)
    ref = res[variants[0]][0]
    for v in variants[1:]:
      got, kern = res[v]
      same = bool(np.array_equal(got, ref))
      f = lambda u: u
      diff = float(np.max(np.abs(f(got) - f(ref)))) if not same else 0.0
      out.append({"parity": [m, n, k], "variant": v, "kernel": kern, "bit_identical": same, "max_abs_diff": diff,
                  "ref_absmax": float(np.max(np.abs(f(ref))))})
      print(json.dumps(out[-1]), flush=True)
    # view kernel, K-contiguous operands, :p7 against auto
    if (m // 256) * (n // 256) >= 192:
      va = _lib.OperandView(m, k, 0, k, 1, 0)
      vb = _lib.OperandView(n, k, 0, k, 1, 0)
      r = {}
      for v in ("auto", "auto:p7"):
        C = DeviceTensor.empty((m, n), _lib.BF16)
        _lib.check(be.lib.tnh_gemm_set_variant(v.encode()))
        try:
          _lib.check(be.lib.tnh_gemm_view(_lib.BF16, _lib.BF16, m, n, k, ctypes.c_void_p(A.ptr), ctypes.byref(va),
                                          ctypes.c_void_p(B.ptr), ctypes.byref(vb), ctypes.c_void_p(C.ptr), n))
          be.synchronize()
        finally:
          _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
        r[v] = np.asarray(C).copy()
      print(json.dumps({"parity_view": [m, n, k], "p7_bit_identical": bool(np.array_equal(r["auto"], r["auto:p7"])),
                        "view_equals_nt": bool(np.array_equal(r["auto"], ref))}), flush=True)
  return out


def headline(be, tel, variants, seconds, d=256):
  m = n = k = d * d
  A = be.device_random((m * k,), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=float(k) ** -0.5)
  B = be.device_random((n * k,), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=float(k) ** -0.5)
  C = DeviceTensor.empty((m, n), _lib.BF16)
  va = _lib.OperandView(m, k, 0, k, 1, 0)
  vb = _lib.OperandView(n, k, 0, k, 1, 0)
  for v in variants:
    plain = v.startswith("plain:")       # the plain NT entry point (tnh_gemm) on the same buffers instead of the view kernel
    _lib.check(be.lib.tnh_gemm_set_variant((v[6:] if plain else v).encode()))

    def call():
      if plain:
        _lib.check(be.lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), k, ctypes.c_void_p(B.ptr), k,
                                   ctypes.c_void_p(C.ptr), n, 1, 0, 0, 0))
        return
      _lib.check(be.lib.tnh_gemm_view(_lib.BF16, _lib.BF16, m, n, k, ctypes.c_void_p(A.ptr), ctypes.byref(va),
                                      ctypes.c_void_p(B.ptr), ctypes.byref(vb), ctypes.c_void_p(C.ptr), n))
    call()
    be.synchronize()
    iters = max(2, int(seconds / 0.4))
    with Sampler(tel) as smp:
      s = _lib.Event().record()
      for _ in range(iters):
        call()
      e = _lib.Event().record()
      e.synchronize()
    _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
    ms = s.elapsed_ms(e) / iters
    tf = 2.0 * m * n * k / ms / 1e9
    rec = {"headline": [m, n, k], "variant": v, "iters": iters, "ms": ms, "tflops": tf, "frac": tf / 2500.0,
           "kernel": be.lib.tnh_gemm_last_kernel().decode()}
    rec.update(smp.summary())
    rec["pj_per_flop"] = (rec["power_mean_w"] / tf) if rec.get("power_mean_w") else None
    print(json.dumps(rec), flush=True)
    time.sleep(0.5)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--seconds", type=float, default=1.5)
  ap.add_argument("--variants", default="auto,bf16_256pp:p7,bf16_256pp:p8,bf16_256pp:p3,bf16_256pp:p6")
  ap.add_argument("--shapes", default="8192x8192x8192,8192x8192x65536")
  ap.add_argument("--fills", default="normal,zeros")
  ap.add_argument("--headline", type=int, default=1)
  ap.add_argument("--headline_variants", default="auto,auto:p7,auto")
  ap.add_argument("--reps", type=int, default=1)
  ap.add_argument("--parity", type=int, default=1)
  a = ap.parse_args()
  be = ta.get_hip_backend()
  tel = Telemetry(be.lib)
  variants = a.variants.split(",")
  if a.parity:
    parity(be, variants)
  for rep in range(a.reps):
    for shape in a.shapes.split(","):
      m, n, k = (int(x) for x in shape.split("x"))
      for fill in a.fills.split(","):
        for v in variants:
          rec = power_probe.run_arm(be, tel, m, n, k, fill, a.seconds, v)
          rec["rep"] = rep
          print(json.dumps(rec), flush=True)
          time.sleep(0.3)
  if a.headline:
    headline(be, tel, a.headline_variants.split(","), 2.5)


if __name__ == "__main__":
  main()
