"""HBM read / copy ceilings of the MI355X (see bw_probe.hip): python tools/bw_probe/run.py   (GPU box)"""
import ctypes
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libbw_probe.so"))
lib.bw_setup.argtypes = [ctypes.c_int64]
lib.bw_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
NAMES = {0: "read", 1: "copy", 2: "copy, non-temporal stores", 3: "hipMemcpyAsync D2D"}
for size in (1 << 30, 64 << 20):
  assert lib.bw_setup(size) == 0
  best = {}
  for mode in (0, 1, 2, 3):
    for unroll in ((1, 2, 4, 8) if mode < 3 else (1,)):
      for grid in ((256 * 4, 256 * 8, 256 * 16, 256 * 32) if mode < 3 else (1,)):
        ms = ctypes.c_float(0)
        assert lib.bw_run(mode, unroll, grid, 10, ctypes.byref(ms)) == 0
        moved = size * (1 if mode == 0 else 2)
        rec = {"bytes": size, "mode": NAMES[mode], "loads_in_flight": unroll, "workgroups": grid, "ms": ms.value,
               "TBps": moved / ms.value / 1e9}
        print(json.dumps(rec), flush=True)
        if mode not in best or rec["TBps"] > best[mode]["TBps"]:
          best[mode] = rec
  for mode, rec in best.items():
    print("BEST", size >> 20, "MiB", NAMES[mode], "%.2f TB/s" % rec["TBps"], "unroll", rec["loads_in_flight"], "grid", rec["workgroups"], flush=True)
