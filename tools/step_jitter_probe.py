"""Per-step wall time, pool footprint and GC activity of the bench's contract_between loop (diagnostic).
  python tools/step_jitter_probe.py [--D 96,256]"""
import argparse, ctypes, gc, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--D", default="96,128,256")
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
be = ta.get_hip_backend()


def footprint():
  u, c, p = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
  _lib.check(be.lib.tnh_mem_stats(ctypes.byref(u), ctypes.byref(c), ctypes.byref(p)))
  return u.value, c.value


for D in [int(x) for x in a.D.split(",")]:
  A, B = bench.make_nodes(ta, be, D, "L0", seed=7, fill="normal")
  for layout in ("L0", "L1"):
    rows = []
    for rep in range(a.reps):
      be.synchronize()
      g0 = gc.get_count()
      t0 = time.perf_counter()
      out = bench.one_step(ta, be, A, B, layout)
      t1 = time.perf_counter()
      del out
      be.synchronize()
      t2 = time.perf_counter()
      u, c = footprint()
      rows.append({"rep": rep, "host_ms": round((t1 - t0) * 1e3, 3), "total_ms": round((t2 - t0) * 1e3, 3),
                   "in_use_GB": round(u / 1e9, 2), "cached_GB": round(c / 1e9, 2), "gc_before": g0})
    print(json.dumps({"D": D, "layout": layout, "ideal_ms": round(2.0 * D**6 / 1.45e15 * 1e3, 3), "steps": rows}), flush=True)
  del A, B
