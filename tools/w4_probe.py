"""A/B of the 4-wave 128x128-wave-tile bf16 GEMM (variant knob ':p6') against the ping-pong kernel.
  python tools/w4_probe.py [--perf]      (GPU box only)"""
import argparse, ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor

ap = argparse.ArgumentParser()
ap.add_argument("--perf", action="store_true")
ap.add_argument("--big", action="store_true")
a = ap.parse_args()
be = ta.get_hip_backend()


def run(variant, code, out_code, A, B, m, n, k, batch=1, iters=0):
  c = DeviceTensor.empty((batch, m, n), out_code)
  _lib.check(be.lib.tnh_gemm_set_variant(variant.encode()))
  def call():
    _lib.check(be.lib.tnh_gemm(code, out_code, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), k, ctypes.c_void_p(B.ptr), k,
                               ctypes.c_void_p(c.ptr), n, batch, m * k, n * k, m * n))
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


ok = True
for (m, n, k, batch) in [] if a.big else [(256, 256, 128, 1), (512, 768, 192, 1), (1024, 1024, 1024, 1), (256, 512, 64 * 7, 3),
                         (2048, 2048, 4096, 1)]:
  for code, dt in ((_lib.BF16, ta.bfloat16), (_lib.F16, np.float16)):
    A = be.device_random((batch * m * k,), dtype=dt, seed=1, normal=True, b=1.0)
    B = be.device_random((batch * n * k,), dtype=dt, seed=2, normal=True, b=1.0)
    for out_code in (code, _lib.F32):
      ref, rname, _ = run("bf16_256pp", code, out_code, A, B, m, n, k, batch)
      r = np.asarray(ref).astype(np.float64)
      for knob in ("p6",):
        got, gname, _ = run("bf16_256pp:" + knob, code, out_code, A, B, m, n, k, batch)
        g = np.asarray(got).astype(np.float64)
        diff = float(np.abs(r - g).max())
        rec = {"m": m, "n": n, "k": k, "batch": batch, "in": code, "out": out_code, "ref_kernel": rname,
               "kernel": gname, "max_abs_diff": diff, "ref_absmax": float(np.abs(r).max())}
        good = "_w4" in gname and diff <= 1e-6 * max(1.0, rec["ref_absmax"])
        rec["ok"] = bool(good)
        ok = ok and good
        print(json.dumps(rec), flush=True)
print("PARITY", "OK" if ok else "FAIL", flush=True)

if a.perf and ok:
  shapes = [(8192, 8192, 8192, 10)]
  if a.big:
    shapes = [(8192, 8192, 65536, 3), (2048, 2048, 262144, 3)]
  for (m, n, k, iters) in shapes:
    for fill in ("random", "zeros"):
      if fill == "zeros":
        A = be.zeros((m * k,), dtype=ta.bfloat16); B = be.zeros((n * k,), dtype=ta.bfloat16)
      else:
        A = be.device_random((m * k,), dtype=ta.bfloat16, seed=1, normal=False, a=-1.0, b=1.0)
        B = be.device_random((n * k,), dtype=ta.bfloat16, seed=2, normal=False, a=-1.0, b=1.0)
      for variant in ("bf16_256pp", "bf16_256pp:p6", "bf16_256pp", "bf16_256pp:p6"):
        _, name, ms = run(variant, _lib.BF16, _lib.BF16, A, B, m, n, k, 1, iters)
        print(json.dumps({"m": m, "n": n, "k": k, "fill": fill, "kernel": name, "ms": ms,
                          "tflops": 2.0 * m * n * k / ms / 1e9}), flush=True)
