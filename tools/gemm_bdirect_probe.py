"""Round-6 A/B of the B-direct loop (knob ":p9": B's MFMA fragments global -> VGPR, no LDS round trip)
against the default lean loop on one box (GPU only).

  python tools/gemm_bdirect_probe.py [--seconds 1.5] [--headline 1]

Step 1 (parity): plain NT entry point and the view kernel, persistent / ragged / short-K shapes -- bit-identical (the
same MFMA sequence).  Step 2 (power): tools/power_probe.run_arm per (shape, variant, fill) with board power and shader
clock.  Step 3: the headline 65536^3 view GEMM, arms alternating.  One JSON line per arm."""
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
from tensornetwork_amd.telemetry import Telemetry  # noqa: E402
import gemm_r5_probe  # noqa: E402
import power_probe  # noqa: E402


def run(be, v, view, m, n, k, A, B, out_dt=None):
  out_dt = _lib.BF16 if out_dt is None else out_dt
  C = DeviceTensor.empty((m, n), out_dt)
  _lib.check(be.lib.tnh_gemm_set_variant(v.encode()))
  try:
    if view:
      va = _lib.OperandView(m, k, 0, k, 1, 0)
      vb = _lib.OperandView(n, k, 0, k, 1, 0)
      _lib.check(be.lib.tnh_gemm_view(_lib.BF16, out_dt, m, n, k, ctypes.c_void_p(A.ptr), ctypes.byref(va),
                                      ctypes.c_void_p(B.ptr), ctypes.byref(vb), ctypes.c_void_p(C.ptr), n))
    else:
      _lib.check(be.lib.tnh_gemm(_lib.BF16, out_dt, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), k, ctypes.c_void_p(B.ptr), k,
                                 ctypes.c_void_p(C.ptr), n, 1, 0, 0, 0))
    be.synchronize()
    kern = be.lib.tnh_gemm_last_kernel().decode()
  finally:
    _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
  return np.asarray(C).copy(), kern


def parity(be):
  ok = True
  for (m, n, k) in [(4096, 4096, 64), (4096, 4096, 128), (4096, 4096, 192), (2560, 2560, 1024), (8192, 8192, 512),
                    (4100, 3972, 1088), (8192, 4096, 4096)]:
    A = be.device_random((m, k), dtype=ta.bfloat16, seed=11, normal=True, a=0.0, b=1.0)
    B = be.device_random((n, k), dtype=ta.bfloat16, seed=12, normal=True, a=0.0, b=1.0)
    for view in (False, True):
      if view and k < 128:
        continue
      for out_dt in (_lib.BF16, _lib.F32):
        try:
          ref, k0 = run(be, "auto" if view else "bf16_256pp", view, m, n, k, A, B, out_dt)
          got, k1 = run(be, "auto:p9" if view else "bf16_256pp:p9", view, m, n, k, A, B, out_dt)
        except Exception as exc:      # a shape the entry point refuses: reported, not fatal
          print(json.dumps({"parity": [m, n, k], "view": view, "error": str(exc)[:120]}), flush=True)
          continue
        same = bool(np.array_equal(got, ref))
        ok = ok and same
        print(json.dumps({"parity": [m, n, k], "view": view, "out_f32": out_dt == _lib.F32, "kernel": k1, "bit_identical": same,
                          "max_abs_diff": 0.0 if same else float(np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64))))}),
              flush=True)
  return ok


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--seconds", type=float, default=1.5)
  ap.add_argument("--shapes", default="8192x8192x8192,8192x8192x65536")
  ap.add_argument("--fills", default="normal,zeros")
  ap.add_argument("--headline", type=int, default=1)
  ap.add_argument("--reps", type=int, default=2)
  a = ap.parse_args()
  be = ta.get_hip_backend()
  tel = Telemetry(be.lib)
  if not parity(be):
    print(json.dumps({"parity_failed": True}), flush=True)
  for rep in range(a.reps):
    for shape in a.shapes.split(","):
      m, n, k = (int(x) for x in shape.split("x"))
      for fill in a.fills.split(","):
        for v in ("bf16_256pp", "bf16_256pp:p9"):
          rec = power_probe.run_arm(be, tel, m, n, k, fill, a.seconds, v)
          rec["rep"] = rep
          print(json.dumps(rec), flush=True)
          time.sleep(0.3)
  if a.headline:
    gemm_r5_probe.headline(be, tel, ["auto", "auto:p9", "auto", "auto:p9"], 2.5)


if __name__ == "__main__":
  main()
