"""Where the time of a ~1 ms contraction step goes: the D = 96 sweep row of bench.py reads 980-1330 TFLOP/s depending
on the box while its GEMM alone runs at ~1450 (tools/tail_probe.py).  One JSON line per variant:

  product            bench.one_step (Node bookkeeping -> contract_between -> tensordot), the allocator's default policy
                     (round 5: collector passes amortised over a bounded growth of the pool)
  product_noslack    the same with the slack switched off: one full pass per step (rounds 1-4)
  tensordot          backend.tensordot on the bare tensors: no Node <-> Edge cycle, so no collector pass is needed
  product_nocollect  no collector pass at all: every dead result waits for Python's own collections (pool growth)

per step: wall ms (best and median of the batches), the host's share (the launch loop returned after ...), the GEMM
kernel's HIP-event time, the collector's counters, tracked Python objects and the cost of one full pass.
`--after-sliced` repeats the product rows after bench.py's sliced-network leg has run in the process (as in the
bench: the sweep comes after it).   python tools/alloc_gc_probe.py [--after-sliced] [D ...]"""
import gc, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tensornetwork_amd as ta
from tensornetwork_amd import _lib, device_tensor as dt
import bench

ta.configure_gc(freeze=True)
be = ta.get_hip_backend()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
dims = [int(a) for a in args] or [96]
after_sliced = "--after-sliced" in sys.argv


def mem():
  import ctypes
  a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
  be.lib.tnh_mem_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
  return {"in_use_mb": a.value >> 20, "cached_mb": b.value >> 20}


def full_pass_ms():
  t0 = time.perf_counter()
  gc.collect()
  return (time.perf_counter() - t0) * 1e3


def measure(name, fn, D, layout, reps=10, batches=5, extra=None):
  t_w = time.perf_counter()
  while time.perf_counter() - t_w < 0.05:
    out = fn(); del out
    be.synchronize()
  m0, s0 = mem(), dt.gc_stats()
  walls, hosts = [], []
  for _ in range(batches):
    t0 = time.perf_counter()
    for _ in range(reps):
      out = fn(); del out
    t1 = time.perf_counter()
    be.synchronize()
    t2 = time.perf_counter()
    walls.append((t2 - t0) / reps * 1e3)
    hosts.append((t1 - t0) / reps * 1e3)
  s1, m1 = dt.gc_stats(), mem()
  be.gemm_events = []
  for _ in range(reps):
    out = fn(); del out
  be.synchronize()
  ev, be.gemm_events = be.gemm_events, None
  gemm_ms = [s.elapsed_ms(e) for s, e in ev]
  flop = 2.0 * D ** 6
  rec = {"variant": name, "D": D, "layout": layout, "wall_ms_best": min(walls), "wall_ms_median": statistics.median(walls),
         "host_ms_best": min(hosts), "host_ms_median": statistics.median(hosts),
         "tflops_best": flop / min(walls) / 1e9, "gemm_event_ms": (sum(gemm_ms) / len(gemm_ms)) if gemm_ms else None,
         "gemm_launches_per_step": len(gemm_ms) / reps, "kernel": be.lib.tnh_gemm_last_kernel().decode(),
         "passes_skipped_for_slack": s1["passes_skipped_for_slack"] - s0["passes_skipped_for_slack"],
         "slack_granted_mb": s1["slack_granted_bytes"] >> 20,
         "full_passes": s1["full_passes"] - s0["full_passes"], "full_pass_estimate_ms": s1["full_pass_seconds"] * 1e3,
         "steps": reps * batches, "pool_before": m0, "pool_after": m1, "tracked_objects": len(gc.get_objects())}
  rec["full_pass_ms_now"] = full_pass_ms()
  if extra:
    rec.update(extra)
  print(json.dumps(rec), flush=True)


def rows(tag):
  for D in dims:
    A, B = bench.make_nodes(ta, be, D, "L0", seed=7, fill="normal")
    for layout in ("L0", "L1"):
      axes = [[2, 3], [0, 1]] if layout == "L0" else [[1, 3], [2, 0]]
      ta.configure_gc(collect_before_large_alloc=True)
      measure("product", lambda: bench.one_step(ta, be, A, B, layout), D, layout, extra={"when": tag})
      gc.collect()
      ta.trim_pool()
      per_size, dt._SLACK_PER_SIZE = dt._SLACK_PER_SIZE, 0
      measure("product_noslack", lambda: bench.one_step(ta, be, A, B, layout), D, layout, extra={"when": tag})
      dt._SLACK_PER_SIZE = per_size
      gc.collect()
      ta.trim_pool()
      measure("tensordot", lambda: be.tensordot(A, B, axes), D, layout, extra={"when": tag})
      if tag == "fresh" and layout == "L0":
        ta.configure_gc(collect_before_large_alloc=False)
        measure("product_nocollect", lambda: bench.one_step(ta, be, A, B, layout), D, layout, batches=2, extra={"when": tag})
        ta.configure_gc(collect_before_large_alloc=True)
        gc.collect()
        ta.trim_pool()
    del A, B
    gc.collect()
    ta.trim_pool()


rows("fresh")
if after_sliced:
  t0 = time.perf_counter()
  rec = bench.sliced_network_bench(ta, be, None, 0, 1, 16, 64, False)
  print(json.dumps({"sliced_leg_seconds": time.perf_counter() - t0, "contraction_seconds": rec.get("seconds"),
                    "tracked_objects": len(gc.get_objects()), "full_pass_ms_now": full_pass_ms()}), flush=True)
  ta.trim_pool()
  rows("after the sliced-network leg")
