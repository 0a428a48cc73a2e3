#!/bin/bash
# chi = 64 MERA placement 'left' for 8 s of its slice loop: the K1 passes through the scalar 64 x 64 tiles
# (TNH_BRICK_RAGGED=0) against the brick kernel (=1); same slices in the same order
for knob in 0 1 0 1; do
  TNH_BRICK_RAGGED=$knob timeout 120 python - <<PY
import json, os, sys
sys.path.insert(0, os.getcwd())
import tensornetwork_amd as ta
from tensornetwork_amd import workloads
ta.configure_gc(freeze=True)
be = ta.get_hip_backend()
run = workloads.mera_sliced_run(be, 64, "left", ta.bfloat16, budget_seconds=8.0, check_every=0)
print(json.dumps({"TNH_BRICK_RAGGED": os.environ["TNH_BRICK_RAGGED"], "slices_done": run["slices_done"], "seconds": run["seconds"],
                  "executed_macs": run.get("executed_macs"), "tflops": 2.0 * run.get("executed_macs", 0.0) / run["seconds"] / 1e12,
                  "permute_launches": be.permute_launches}))
PY
done
