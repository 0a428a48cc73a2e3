"""Strong-scaling emulation on ONE GPU: time rank 0's share of the sliced 64-node network for
world = 1, 2, 4, 8 (the other ranks' shares are identical in size) -> predicted N-GPU speed-up."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import distributed, workloads
D = int(sys.argv[1]) if len(sys.argv) > 1 else 12
be = ta.get_hip_backend()
tensors = workloads.random_regular_device_tensors(be, 64, D, ta.bfloat16, seed=6)
nodes = workloads.random_regular_network(be, n=64, D=D, seed=6, tensors=tensors)
cuts = distributed.choose_cut_edges(nodes, min_slices=64)
rep = distributed.slicing_report(nodes, cuts)
class Sub(distributed.LocalComm):
  def __init__(self, world): self.rank, self.world = 0, world
distributed.contract_sliced(nodes, cuts, comm=Sub(int(rep["n_slices"]) // 2)); be.synchronize()
base = None
for world in (1, 2, 4, 8):
  be.synchronize(); t0 = time.perf_counter()
  out = distributed.contract_sliced(nodes, cuts, comm=Sub(world)); be.synchronize()
  t = time.perf_counter() - t0
  base = base or t
  print(json.dumps({"D": D, "world": world, "slices": len(range(0, int(rep["n_slices"]), world)), "seconds": t,
                    "speedup_vs_1": base / t}), flush=True)
