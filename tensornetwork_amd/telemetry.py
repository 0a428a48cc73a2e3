"""Board power and shader clock of THIS process' GPU, from sysfs (amdgpu hwmon).

The headline GEMM is power-bound on random operands: the chip lowers its clock
to stay inside the board power cap, so "fraction of the 2.5 PFLOP/s peak" mixes kernel quality with
DVFS.  ``Telemetry`` finds the hwmon directory of the device libtnhip is bound to (by PCI address,
``tnh_device_pci_bus_id``) and ``Sampler`` polls it from a host thread while a timed region runs, so
that bench.py can print the achieved fraction of the matrix-core peak AT THE OBSERVED CLOCK next to
the plain fraction.  Read-only files; nothing here touches the device."""
import ctypes
import glob
import os
import re
import threading
import time


def _read(path):
  try:
    with open(path) as f:
      return f.read().strip()
  except OSError:
    return None


class Telemetry:
  """Sensors of one AMD GPU.  ``sample()`` -> (watts or None, shader MHz or None)."""

  def __init__(self, lib=None, pci=None):
    if pci is None and lib is not None:
      buf = ctypes.create_string_buffer(64)
      if lib.tnh_device_pci_bus_id(buf, 64) == 0:
        pci = buf.value.decode().lower()
    self.pci = pci
    self.card = None
    self.power_file = self.cap_file = self.freq_file = self.sclk_file = None
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    chosen = None
    for dev in cards:
      if _read(os.path.join(dev, "vendor")) != "0x1002":
        continue
      addr = os.path.basename(os.path.realpath(dev)).lower()
      if pci is None or addr == pci:
        chosen = dev
        break
    if chosen is None:
      return
    self.card = chosen
    for hw in sorted(glob.glob(os.path.join(chosen, "hwmon", "hwmon*"))):
      for name in ("power1_average", "power1_input"):
        if self.power_file is None and _read(os.path.join(hw, name)):
          self.power_file = os.path.join(hw, name)
      if self.cap_file is None and _read(os.path.join(hw, "power1_cap")):
        self.cap_file = os.path.join(hw, "power1_cap")
      if self.freq_file is None and _read(os.path.join(hw, "freq1_input")):
        self.freq_file = os.path.join(hw, "freq1_input")
    if os.path.exists(os.path.join(chosen, "pp_dpm_sclk")):
      self.sclk_file = os.path.join(chosen, "pp_dpm_sclk")

  @property
  def available(self):
    return self.power_file is not None or self.freq_file is not None or self.sclk_file is not None

  def describe(self):
    return {"pci": self.pci, "card": self.card, "power": self.power_file, "cap": self.cap_file,
            "freq": self.freq_file, "sclk": self.sclk_file}

  def cap_watts(self):
    v = _read(self.cap_file) if self.cap_file else None
    return float(v) / 1e6 if v else None

  def sample(self):
    watts = mhz = None
    v = _read(self.power_file) if self.power_file else None
    if v:
      watts = float(v) / 1e6
    v = _read(self.freq_file) if self.freq_file else None
    if v:
      mhz = float(v) / 1e6
    if mhz is None and self.sclk_file:
      m = re.search(r"(\d+)\s*Mhz\s*\*", _read(self.sclk_file) or "", flags=re.I)
      mhz = float(m.group(1)) if m else None
    return watts, mhz


class Sampler:
  """``with Sampler(tel) as s: ...`` polls ``tel`` every ``period`` s; ``s.summary()`` afterwards."""

  def __init__(self, tel, period=0.02, skip_fraction=0.25):
    self.tel, self.period, self.skip = tel, period, skip_fraction
    self.samples = []
    self._stop = threading.Event()
    self._thread = None

  def __enter__(self):
    if self.tel is not None and self.tel.available:
      self._thread = threading.Thread(target=self._poll, daemon=True)
      self._thread.start()
    return self

  def _poll(self):
    while not self._stop.is_set():
      self.samples.append(self.tel.sample())
      time.sleep(self.period)

  def __exit__(self, *exc):
    self._stop.set()
    if self._thread is not None:
      self._thread.join()
    return False

  def summary(self):
    body = self.samples[int(len(self.samples) * self.skip):] or self.samples   # drop the DVFS ramp
    watts = [w for w, _ in body if w is not None]
    mhz = [c for _, c in body if c is not None]
    mean = lambda xs: (sum(xs) / len(xs)) if xs else None
    return {"n_samples": len(body), "power_mean_w": mean(watts), "power_max_w": max(watts) if watts else None,
            "power_cap_w": self.tel.cap_watts() if self.tel is not None else None,
            "sclk_mean_mhz": mean(mhz), "sclk_min_mhz": min(mhz) if mhz else None}
