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


def _read(path):
  try:
    with open(path) as f:
      return f.read().strip()
  except OSError:
    return None


class Telemetry:
  """Finds the sensors of the first AMD GPU once; ``sample()`` returns (watts, sclk_mhz) or Nones."""

  def __init__(self):
    self.power_file = self.cap_file = self.sclk_file = self.freq_file = None
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
      if _read(os.path.join(dev, "vendor")) != "0x1002":
        continue
      for hw in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))):
        for name in ("power1_average", "power1_input"):
          if self.power_file is None and _read(os.path.join(hw, name)) not in (None, ""):
            self.power_file = os.path.join(hw, name)
        if self.cap_file is None and _read(os.path.join(hw, "power1_cap")):
          self.cap_file = os.path.join(hw, "power1_cap")
        if self.freq_file is None and _read(os.path.join(hw, "freq1_input")):
          self.freq_file = os.path.join(hw, "freq1_input")
      if os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
        self.sclk_file = os.path.join(dev, "pp_dpm_sclk")
      if self.power_file:
        break
    self.source = {"power": self.power_file, "cap": self.cap_file, "sclk": self.sclk_file, "freq": self.freq_file}
    self.smi = None
    if self.power_file is None:
      for tool in ("/opt/rocm/bin/amd-smi", "/opt/rocm/bin/rocm-smi"):
        if os.path.exists(tool):
          self.smi = tool
          break
      self.source["smi"] = self.smi

  def cap_watts(self):
    v = _read(self.cap_file) if self.cap_file else None
    return float(v) / 1e6 if v else None

  def sample(self):
    watts = mhz = None
    if self.power_file:
      v = _read(self.power_file)
      watts = float(v) / 1e6 if v else None
    if self.freq_file:
      v = _read(self.freq_file)
      mhz = float(v) / 1e6 if v else None
    if mhz is None and self.sclk_file:
      txt = _read(self.sclk_file) or ""
      m = re.search(r"(\d+)\s*Mhz\s*\*", txt, flags=re.I)
      mhz = float(m.group(1)) if m else None
    if watts is None and self.smi:
      watts, mhz2 = self._smi_sample()
      mhz = mhz if mhz is not None else mhz2
    return watts, mhz

  def _smi_sample(self):
    try:
      if self.smi.endswith("amd-smi"):
        out = subprocess.run([self.smi, "metric", "-g", "0", "-p", "-c", "--json"], capture_output=True, text=True,
                             timeout=5, check=False).stdout
        w = re.search(r'"socket_power"\s*:\s*\{[^}]*?"value"\s*:\s*([\d.]+)', out, flags=re.S)
        c = re.search(r'"gfx_0"\s*:\s*\{[^}]*?"clk"\s*:\s*\{[^}]*?"value"\s*:\s*([\d.]+)', out, flags=re.S)
        return (float(w.group(1)) if w else None), (float(c.group(1)) if c else None)
      out = subprocess.run([self.smi, "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                           timeout=5, check=False).stdout
      w = re.search(r'Power \(W\)"\s*:\s*"([\d.]+)"', out)
      c = re.search(r'sclk clock speed:"\s*:\s*"\((\d+)Mhz\)"', out)
      return (float(w.group(1)) if w else None), (float(c.group(1)) if c else None)
    except Exception:  # pylint: disable=broad-except
      return None, None


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
  samples, stop = [], threading.Event()

  def poll():
    while not stop.is_set():
      samples.append(tel.sample())
      time.sleep(0.02)
  th = threading.Thread(target=poll, daemon=True)
  s = _lib.Event().record()
  th.start()
  for _ in range(iters):
    call()
  e = _lib.Event().record()
  e.synchronize()
  stop.set(); th.join()
  _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
  ms = s.elapsed_ms(e) / iters
  # drop the ramp: first 25 % of the samples
  body = samples[len(samples) // 4:] or samples
  watts = [w for w, _ in body if w is not None]
  mhz = [c for _, c in body if c is not None]
  tf = 2.0 * m * n * k / ms / 1e9
  clock = float(np.mean(mhz)) if mhz else None
  rec = {"m": m, "n": n, "k": k, "fill": fill, "variant": variant, "iters": iters, "ms": ms, "tflops": tf,
         "kernel": be.lib.tnh_gemm_last_kernel().decode(),
         "power_mean_w": float(np.mean(watts)) if watts else None, "power_max_w": float(np.max(watts)) if watts else None,
         "power_cap_w": tel.cap_watts(), "sclk_mean_mhz": clock, "sclk_min_mhz": float(np.min(mhz)) if mhz else None,
         "n_samples": len(body),
         "mfma_frac_of_2p5pf": tf / 2500.0,
         "mfma_frac_at_clock": (tf / (2500.0 * clock / 2400.0)) if clock else None}
  del A, B, C
  return rec


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--seconds", type=float, default=2.5)
  ap.add_argument("--shapes", default="8192x8192x8192,8192x8192x65536,8192x8192x262144")
  ap.add_argument("--fills", default="zeros,ones,normal,uniform")
  ap.add_argument("--variants", default="auto")
  a = ap.parse_args()
  be = ta.get_hip_backend()
  be.lib  # pylint: disable=pointless-statement
  tel = Telemetry()
  print(json.dumps({"telemetry_sources": tel.source, "idle_sample": tel.sample(), "cap_w": tel.cap_watts()}), flush=True)
  for shape in a.shapes.split(","):
    m, n, k = (int(x) for x in shape.split("x"))
    for variant in a.variants.split(","):
      for fill in a.fills.split(","):
        print(json.dumps(run_arm(be, tel, m, n, k, fill, a.seconds, variant)), flush=True)
        time.sleep(0.5)


if __name__ == "__main__":
  main()
