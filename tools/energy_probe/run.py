"""Energy decomposition of the bf16 GEMM on the MI355X (see energy_probe.hip): python tools/energy_probe/run.py"""
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import tensornetwork_amd as ta  # noqa: E402
from tensornetwork_amd.telemetry import Sampler, Telemetry  # noqa: E402

be = ta.get_hip_backend()
be.lib  # pylint: disable=pointless-statement
tel = Telemetry(be.lib)
lib = ctypes.CDLL(os.path.join(HERE, "libenergy_probe.so"))
lib.probe_setup.argtypes = [ctypes.c_int64]
lib.probe_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
NAMES = {0: "MFMA only (operands in registers)", 1: "+ LDS fragment reads (24 ds_read_b128 / 64 MFMA)",
         2: "+ global -> LDS traffic, cache resident (32 MiB)", 3: "+ global -> LDS traffic from HBM (16 GiB)"}
GRID = 256
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
print(json.dumps({"cap_w": tel.cap_watts(), "idle": tel.sample()}), flush=True)
for mode in (0, 1, 2, 3):
  rc = lib.probe_setup((16 << 30) if mode == 3 else (32 << 20))
  assert rc == 0, rc
  ms = ctypes.c_float(0)
  ktiles = 4096
  assert lib.probe_run(mode, GRID, ktiles, 2, ctypes.byref(ms)) == 0
  launches = max(3, int(seconds * 1e3 / max(ms.value, 1e-3)))
  with Sampler(tel) as smp:
    assert lib.probe_run(mode, GRID, ktiles, launches, ctypes.byref(ms)) == 0
  flops = GRID * 8.0 * ktiles * 32 * 16384
  tf = flops / ms.value / 1e9
  rec = {"mode": mode, "what": NAMES[mode], "ms_per_launch": ms.value, "tflops": tf,
         "global_GBps": (GRID * ktiles * 65536 / ms.value / 1e6) if mode >= 2 else 0.0}
  rec.update(smp.summary())
  rec["pj_per_flop"] = rec["power_mean_w"] / tf if rec.get("power_mean_w") else None
  print(json.dumps(rec), flush=True)
