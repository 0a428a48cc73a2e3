"""One-GPU rehearsal of the 8-rank run of ONE placement of the chi = 64 bond-sliced MERA layer (GPU box only).

  python tools/mera_rank_share_rehearsal.py [chi=64] [world=8] [placement=left] [one_rank=1]

The chi^2 slices of `workloads.MeraSlicedLayer` are dealt to `world` ranks by `_StagePlan.partition` (the rule
`contract_sliced` applies: contiguous blocks, the loop order and block rule whose slowest rank has the smallest
estimated time).  Every rank's share is contracted on this one GPU, one after the other, through
`distributed._contract_slices_staged` (steps once per distinct value of the cuts they depend on -- a rank repeats the
single-cut classes its block touches), timed separately.  Reported: seconds and executed multiply-adds per rank,
the model's multiply-adds for the same block, the slowest rank, the sum of the ranks' partial energies against the
one-rank energy (the all-reduce is a scalar sum) and, with one_rank=1, the one-GPU run of all slices on the same box ->
predicted speed-up of the 8-rank run.  Nothing is emulated but the communicator."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta  # noqa: E402
from tensornetwork_amd import workloads  # noqa: E402

CHI = int(sys.argv[1]) if len(sys.argv) > 1 else 64
WORLD = int(sys.argv[2]) if len(sys.argv) > 2 else 8
PLACEMENT = sys.argv[3] if len(sys.argv) > 3 else "left"
ONE_RANK = int(sys.argv[4]) if len(sys.argv) > 4 else 1
be = ta.get_hip_backend()
layer = workloads.MeraSlicedLayer(be, CHI, PLACEMENT, ta.bfloat16, seed=40)
plan = layer.stage
slices = layer.all_slices()
blocks = plan.partition(slices, WORLD)
layer.contract(slices[:2])            # warm-up: kernels, allocator, the shared tensors
be.synchronize()


def run(block):
  st, marks = {}, []
  be.synchronize()
  t0 = time.perf_counter()

  def on_slice(done, idx, tensor):      # pylint: disable=unused-argument
    if done in (1, len(block) // 8, len(block) // 2):
      be.synchronize()
      marks.append([done, time.perf_counter() - t0])
    return False
  acc = layer.contract(block, stats=st, on_slice=on_slice)
  be.synchronize()
  st["seconds_at_slices_done"] = marks
  return time.perf_counter() - t0, float(np.asarray(acc, dtype=np.float64).reshape(-1)[0]), st


rows = []
for rank, block in enumerate(blocks):
  sec, part, st = run(block)
  rows.append({"rank": rank, "slices": len(block), "first": list(block[0]) if block else None,
               "last": list(block[-1]) if block else None, "seconds": sec, "partial_energy": part,
               "executed_macs": float(st["executed_macs"]), "model_macs": float(plan.macs_with_reuse(block)),
               "model_seconds": float(plan.seconds_with_reuse(block)), "stage_runs": st["stage_runs"],
               "seconds_at_slices_done": st["seconds_at_slices_done"],
               "kept_for_all_values": st.get("classes_kept_for_all_values")})
  print(json.dumps(rows[-1]), flush=True)
slowest = max(r["seconds"] for r in rows)
total = float(sum(r["partial_energy"] for r in rows))
summary = {"chi": CHI, "world": WORLD, "placement": PLACEMENT, "slowest_rank_seconds": slowest,
           "sum_of_rank_seconds": float(sum(r["seconds"] for r in rows)), "sum_of_rank_partials": total,
           "executed_macs_all_ranks": float(sum(r["executed_macs"] for r in rows)),
           "executed_equals_model_on_every_rank": bool(all(abs(r["executed_macs"] - r["model_macs"]) <= 1e-9 * r["model_macs"]
                                                           for r in rows)),
           "model_macs_one_rank": float(plan.macs_with_reuse(slices)),
           "model_ideal_speedup": float(plan.macs_with_reuse(slices) / max(plan.macs_with_reuse(b) for b in blocks))}
if ONE_RANK:
  sec1, e1, st1 = run(slices)
  summary.update({"one_rank_seconds": sec1, "one_rank_energy": e1, "one_rank_executed_macs": float(st1["executed_macs"]),
                  "predicted_speedup": sec1 / slowest,
                  "rel_diff_sum_of_partials_vs_one_rank": abs(total - e1) / max(abs(e1), 1e-30),
                  "note": "f32 sums of bf16-rounded slice energies in different groupings: agreement to f32 rounding of "
                          "the running sum, not bit-equality"})
print(json.dumps(summary), flush=True)
