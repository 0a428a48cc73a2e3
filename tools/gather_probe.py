"""Gather GEMM against permute + streaming GEMM on the operand shapes of the D = 12 network: a 144 x 144 tensor
takes two bonds off a rank-9 intermediate (12^8 = 430 M elements) whose contracted axes sit at different places.
  python tools/gather_probe.py [--reps 5]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import hip_backend

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--cases", default="")
a = ap.parse_args()
be = ta.get_hip_backend()
D = 12
long_shape = (D,) * 8
small = be.device_random((D, D, D, D), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=0.1)
long_ = be.device_random(long_shape, dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=0.1)
nbytes = 2 * (12**8 + 144 * 12**6 + 144 * 144)          # algorithmic: read the long operand, write the result

CASES = {            # contracted axes of the long tensor
    "k37_run12": [3, 7],        # innermost axis contracted, free run of 1728 below it: BN 64, pieces 1536 B
    "k47_free144": [4, 7],      # innermost contracted, free run of 144: BN 48
    "k15_inner_free12": [1, 5],  # innermost axes (6, 7) free: 144 -> BN 48, pieces 96 B
    "k16_inner_free12": [1, 6],  # innermost axis free (12), contracted axis right above: BN 48, pieces 1152 B
    "k01_kmajor": [0, 1],       # k-major: BN 64, pieces of 128 B
    "k67_trailing": [6, 7],     # contracted axes trailing: never gathered (the streaming kernel's own case)
}


def timed(fn):
  out = fn(); del out
  be.synchronize()
  best = 1e9
  for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(a.reps):
      out = fn(); del out
    be.synchronize()
    best = min(best, (time.perf_counter() - t0) / a.reps)
  return best


want = [c for c in a.cases.split(",") if c] or list(CASES)
for name in want:
  axes_l = CASES[name]
  plan = hip_backend._gather_descriptor(long_shape, axes_l)
  info = None if plan is None else {"bn": plan[1], "piece_bytes": hip_backend._gather_piece_bytes(plan[0]),
                                    "innermost_contracted": bool(plan[0].k_mask & 1)}
  rec = {"case": name, "axes": axes_l, "plan": info}
  for orient in ("small_first", "long_first"):
    args = (small, long_, [[1, 3], axes_l]) if orient == "small_first" else (long_, small, [axes_l, [1, 3]])
    for mode in ("classic", "gather"):
      be.gather_gemm = mode == "gather"
      g0, p0 = be.gather_launches, be.permute_launches
      t = timed(lambda: be.tensordot(*args))
      rec[f"{orient}_{mode}"] = {"us": round(t * 1e6, 1), "TB/s": round(nbytes / t / 1e12, 2),
                                 "gather_launches": (be.gather_launches - g0) // (3 * a.reps + 1),
                                 "permutes": (be.permute_launches - p0) // (3 * a.reps + 1),
                                 "kernel": be.lib.tnh_gemm_last_kernel().decode()}
  print(json.dumps(rec), flush=True)
