"""One-GPU rehearsal of the strong-scaling run of the sliced 64-node network (GPU box only).

  python tools/rr_scaling_rehearsal.py [D=12] [reps=3]

For both cut choices -- `beam=0` (the sequential rule of rounds 1-4) and the default of round 5 (cut SETS ranked by
the slowest rank's executed multiply-adds) -- and world = 1, 2, 4, 8: EVERY rank's share (contract_sliced's own
partition and mode decision, nothing emulated but the communicator) is contracted on this one GPU, one rank after the
other, timed separately (best of `reps`, per-rank fixed costs included: the slice-invariant steps run once per rank).
Reported per (cuts, world): the slowest rank's seconds, the predicted speed-up over the same cuts' one-GPU time and
over the BEST one-GPU time of either cut choice, the model's figure (slicing_report), the sum of the ranks' partial
results against the one-rank result (the all-reduce is a sum) and, once per cut choice, against f32 on the same
bf16-rounded tensors."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta  # noqa: E402
from tensornetwork_amd import distributed, workloads  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 12
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
be = ta.get_hip_backend()
tensors = workloads.random_regular_device_tensors(be, 64, D, ta.bfloat16, seed=6)
nodes = workloads.random_regular_network(be, n=64, D=D, seed=6, tensors=tensors)


class Sub(distributed.LocalComm):
  def __init__(self, rank, world):
    self.rank, self.world = rank, world


one_gpu = {}
rows = []
for label, kw in (("sequential (rounds 1-4)", {"beam": 0}), ("slowest-rank beam (round 5 default)", {"world": 8})):
  t0 = time.perf_counter()
  cuts = distributed.choose_cut_edges(nodes, min_slices=64, **kw)
  t_search = time.perf_counter() - t0
  model1 = distributed.slicing_report(nodes, cuts, world=1)
  distributed.contract_sliced(nodes, cuts, comm=Sub(0, 2))      # warm-up (allocator, kernels)
  be.synchronize()
  ref_sum = None
  for world in (1, 2, 4, 8):
    model = distributed.slicing_report(nodes, cuts, world=world)
    secs, parts, executed, modes = [], [], [], set()
    for rank in range(world):
      best = None
      for _ in range(REPS):
        st = {}
        be.synchronize()
        t0 = time.perf_counter()
        out = distributed.contract_sliced(nodes, cuts, comm=Sub(rank, world), stats=st)
        be.synchronize()
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
      secs.append(best)
      parts.append(float(np.asarray(out, dtype=np.float64).reshape(-1)[0]))
      executed.append(st.get("executed_macs"))
      modes.add(st.get("mode"))
    total = float(sum(parts))
    if world == 1:
      ref_sum = total
      one_gpu[label] = secs[0]
    rows.append({"cuts": label, "D": D, "world": world, "cut_search_seconds": t_search, "modes": sorted(modes),
                 "slices_per_rank": model.get("slices_per_rank"), "seconds_per_rank": secs, "slowest_rank_seconds": max(secs),
                 "speedup_vs_same_cuts_1gpu": one_gpu[label] / max(secs),
                 "model_ideal_speedup": model1["flops_with_reuse_slowest_rank"] / model["flops_with_reuse_slowest_rank"],
                 "executed_macs_slowest": max(executed), "model_macs_slowest": model["flops_with_reuse_slowest_rank"],
                 "sum_of_rank_partials": total, "one_rank_result": ref_sum,
                 "rel_diff_vs_one_rank": abs(total - ref_sum) / max(abs(ref_sum), 1e-30)})
    print(json.dumps(rows[-1]), flush=True)
  # f32 on the same (bf16-valued) tensors, same cuts
  t32 = [be.cast(x, np.float32) for x in tensors]
  nodes32 = workloads.random_regular_network(be, n=64, D=D, seed=6, tensors=t32)
  cuts32 = distributed.choose_cut_edges(nodes32, min_slices=64, **kw)
  r32 = float(np.asarray(distributed.contract_sliced(nodes32, cuts32), dtype=np.float64).reshape(-1)[0])
  print(json.dumps({"cuts": label, "bf16_one_rank": ref_sum, "f32_same_cuts": r32,
                    "rel_err_of_the_sum": abs(ref_sum - r32) / max(abs(r32), 1e-30)}), flush=True)
  del t32, nodes32
best1 = min(one_gpu.values())
for r in rows:
  if r["world"] > 1:
    print(json.dumps({"cuts": r["cuts"], "world": r["world"], "slowest_rank_seconds": r["slowest_rank_seconds"],
                      "speedup_vs_best_1gpu_of_either": best1 / r["slowest_rank_seconds"], "best_1gpu_seconds": best1}), flush=True)
