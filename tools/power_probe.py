"""Board power and shader clock while the headline bf16 GEMM runs (GPU box only).

  python tools/power_probe.py [--seconds 2.5] [--shapes 8192x8192x8192,8192x8192x65536,...]

For every (shape, operand fill) the ping-pong kernel is launched back to back for ~`seconds`
while a host thread samples the GPU's power sensor and current shader clock (sysfs hwmon /
pp_dpm_sclk first; `amd-smi` / `rocm-smi` snapshots as fall-backs, whatever the box has).
One JSON line per arm: TFLOP/s, mean / max power, the power cap, the mean shader clock, and
``mfma_frac_at_clock`` = achieved TFLOP/s / (2.5 PFLOP/s x mean_clock / 2.4 GHz) -- the fraction
of the matrix-pipe peak AT THE CLOCK THE CHIP ACTUALLY RAN.  The claim under test (DESIGN.md
section 4): on random operands the kernel is power-bound -- power sits at the cap and the clock
drops, while on zero-filled operands the same binary clocks near 2.4 GHz."""
import argparse
import ctypes
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta  # noqa: E402
from tensornetwork_amd import _lib  # noqa: E402
from tensornetwork_amd.device_tensor import DeviceTensor  # noqa: E402


from tensornetwork_amd.telemetry import Sampler, Telemetry  # noqa: E402


DT = {"bf16": (ta.bfloat16, _lib.BF16, 2500.0), "f32": (np.float32, _lib.F32, 157.3), "f64": (np.float64, _lib.F64, 78.6)}


def run_arm_dtype(be, tel, m, n, k, fill, seconds, dtype):
  """f32 / f64 arms: the native matrix-instruction kernels (':s0' keeps f32 off the 3 x bf16 path)."""
  npdt, code, peak = DT[dtype]
  if fill == "zeros":
    A, B = be.zeros((m * k,), dtype=npdt), be.zeros((n * k,), dtype=npdt)
  else:
    A = be.device_random((m * k,), dtype=npdt, seed=1, normal=True, a=0.0, b=float(k) ** -0.5)
    B = be.device_random((n * k,), dtype=npdt, seed=2, normal=True, a=0.0, b=float(k) ** -0.5)
  C = DeviceTensor.empty((m, n), code)
  _lib.check(be.lib.tnh_gemm_set_variant(b"auto:s0"))

  def call():
    _lib.check(be.lib.tnh_gemm(code, code, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), k, ctypes.c_void_p(B.ptr), k,
                               ctypes.c_void_p(C.ptr), n, 1, 0, 0, 0))
  call()
  be.synchronize()
  s0 = _lib.Event().record(); call(); e0 = _lib.Event().record(); e0.synchronize()
  iters = max(3, int(seconds * 1e3 / max(s0.elapsed_ms(e0), 1e-3)))
  with Sampler(tel) as smp:
    s = _lib.Event().record()
    for _ in range(iters):
      call()
    e = _lib.Event().record()
    e.synchronize()
  _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
  ms = s.elapsed_ms(e) / iters
  tf = 2.0 * m * n * k / ms / 1e9
  rec = {"dtype": dtype, "m": m, "n": n, "k": k, "fill": fill, "iters": iters, "ms": ms, "tflops": tf,
         "kernel": be.lib.tnh_gemm_last_kernel().decode(), "peak_tflops": peak}
  rec.update(smp.summary())
  clock = rec["sclk_mean_mhz"]
  rec["frac_of_peak"] = tf / peak
  rec["frac_at_clock"] = (tf / (peak * clock / 2400.0)) if clock else None
  return rec


def run_arm(be, tel, m, n, k, fill, seconds, variant):
  if fill == "zeros":
    A, B = be.zeros((m * k,), dtype=ta.bfloat16), be.zeros((n * k,), dtype=ta.bfloat16)
  elif fill == "normal":
    s = float(k) ** -0.5
    A = be.device_random((m * k,), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=s)
    B = be.device_random((n * k,), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=s)
  elif fill == "uniform":
    A = be.device_random((m * k,), dtype=ta.bfloat16, seed=1, normal=False, a=-1.0, b=1.0)
    B = be.device_random((n * k,), dtype=ta.bfloat16, seed=2, normal=False, a=-1.0, b=1.0)
  elif fill == "ones":
    A, B = be.ones((m * k,), dtype=ta.bfloat16), be.ones((n * k,), dtype=ta.bfloat16)
  else:
    raise ValueError(fill)
  C = DeviceTensor.empty((m, n), _lib.BF16)
  _lib.check(be.lib.tnh_gemm_set_variant(variant.encode()))

  def call():
    _lib.check(be.lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), k,
                               ctypes.c_void_p(B.ptr), k, ctypes.c_void_p(C.ptr), n, 1, 0, 0, 0))
  call()
  be.synchronize()
  s0 = _lib.Event().record(); call(); e0 = _lib.Event().record(); e0.synchronize()
  one = max(s0.elapsed_ms(e0), 1e-3)
  iters = max(3, int(seconds * 1e3 / one))
  with Sampler(tel) as smp:
    s = _lib.Event().record()
    for _ in range(iters):
      call()
    e = _lib.Event().record()
    e.synchronize()
  _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
  ms = s.elapsed_ms(e) / iters
  tf = 2.0 * m * n * k / ms / 1e9
  rec = {"m": m, "n": n, "k": k, "fill": fill, "variant": variant, "iters": iters, "ms": ms, "tflops": tf,
         "kernel": be.lib.tnh_gemm_last_kernel().decode()}
  rec.update(smp.summary())
  clock = rec["sclk_mean_mhz"]
  rec["pj_per_flop"] = (rec["power_mean_w"] / tf) if rec.get("power_mean_w") else None      # W / (TFLOP/s) = pJ / flop
  rec["mfma_frac_of_2p5pf"] = tf / 2500.0
  rec["mfma_frac_at_clock"] = (tf / (2500.0 * clock / 2400.0)) if clock else None
  del A, B, C
  return rec


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--seconds", type=float, default=2.5)
  ap.add_argument("--shapes", default="8192x8192x8192,8192x8192x65536,8192x8192x262144")
  ap.add_argument("--fills", default="zeros,ones,normal,uniform")
  ap.add_argument("--variants", default="auto")
  ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f64"])
  a = ap.parse_args()
  be = ta.get_hip_backend()
  be.lib  # pylint: disable=pointless-statement
  tel = Telemetry(be.lib)
  print(json.dumps({"telemetry_sources": tel.describe(), "idle_sample": tel.sample(), "cap_w": tel.cap_watts()}), flush=True)
  for shape in a.shapes.split(","):
    m, n, k = (int(x) for x in shape.split("x"))
    for variant in a.variants.split(","):
      for fill in a.fills.split(","):
        if a.dtype != "bf16":
          print(json.dumps(run_arm_dtype(be, tel, m, n, k, fill, a.seconds, a.dtype)), flush=True)
        else:
          print(json.dumps(run_arm(be, tel, m, n, k, fill, a.seconds, variant)), flush=True)
        time.sleep(0.5)


if __name__ == "__main__":
  main()
