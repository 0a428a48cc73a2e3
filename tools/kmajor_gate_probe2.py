"""Below the 2 GiB gate: k-major operand read in place (nn view kernel) against ONE K1 permute + the NT view kernel,
config-2 L0 at D = 64 / 96 / 128 (b = 34 / 170 / 537 MB) with the round-5 lean loops.  GPU box only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
import bench
be = ta.get_hip_backend()
default_gate = be.inplace_max_bytes
for D in (64, 96, 128):
  A, B = bench.make_nodes(ta, be, D, "L0", seed=7, fill="normal")
  for rep in range(2):
    for gate in (default_gate, 0):
      be.inplace_max_bytes = gate
      be.kmajor_inplace_penalty = 0.0        # the size gate alone decides: in place (default gate) or ONE K1 pass (gate 0)
      t, permutes = bench.timed_steps(be, lambda: bench.one_step(ta, be, A, B, "L0"), 10, batches=3)
      print(json.dumps({"case": f"D{D}_L0", "gate_bytes": gate, "rep": rep, "ms": t * 1e3, "tflops": 2.0 * D**6 / t / 1e12,
                        "permute_launches": permutes, "kernel": be.lib.tnh_gemm_last_kernel().decode()}), flush=True)
  be.inplace_max_bytes = default_gate
  del A, B
