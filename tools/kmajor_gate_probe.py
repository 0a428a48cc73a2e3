"""The 2 GiB gate on k-major operands read in place (HipBackend.inplace_max_bytes) with the round-5 lean loop:
config-2 L0 at D = 192 / 256 (b is [K][N]: 2.7 / 8.6 GB) and the D = 512 row (4.3 GB), whole contract_between step,
gate at 2 GiB (permute + NT) against no gate (in place).  GPU box only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
import bench
be = ta.get_hip_backend()
default_gate = be.inplace_max_bytes


def row(name, step, flops):
  for rep in range(2):
    for gate in (default_gate, 1 << 42):
      be.inplace_max_bytes = gate
      be.kmajor_inplace_penalty = 0.0        # the size gate alone decides
      t, permutes = bench.timed_steps(be, step, 3, batches=1)
      print(json.dumps({"case": name, "gate_bytes": gate, "rep": rep, "ms": t * 1e3, "tflops": flops / t / 1e12,
                        "permute_launches": permutes, "kernel": be.lib.tnh_gemm_last_kernel().decode()}), flush=True)
  be.inplace_max_bytes = default_gate


for D in (192, 256):
  A, B = bench.make_nodes(ta, be, D, "L0", seed=7, fill="normal")
  row(f"D{D}_L0", lambda: bench.one_step(ta, be, A, B, "L0"), 2.0 * D**6)
  del A, B
A = be.device_random((64, 128, 512, 512), dtype=ta.bfloat16, seed=11, normal=True, a=0.0, b=1.0 / 512)
B = be.device_random((512, 512, 128, 64), dtype=ta.bfloat16, seed=12, normal=True, a=0.0, b=1.0 / 512)


def step512():
  a, b = ta.Node(A, backend=be), ta.Node(B, backend=be)
  a[2] ^ b[0]  # pylint: disable=pointless-statement
  a[3] ^ b[1]  # pylint: disable=pointless-statement
  return ta.contract_between(a, b)


row("D512row", step512, 2.0 * 8192 * 8192 * 262144)
