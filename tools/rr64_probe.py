"""North-star scaling network on one GPU: 64-node random 3-regular network, bond D,
contracted by bond slicing (tensornetwork_amd.distributed).  Reports per-slice time.
  python tools/rr64_probe.py --D 8 --min-slices 64 --max-slices 8 [--dtype bf16|f32]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import distributed, network, contractors, pathfinder

ap = argparse.ArgumentParser()
ap.add_argument("--D", type=int, default=8)
ap.add_argument("--n", type=int, default=64)
ap.add_argument("--min-slices", type=int, default=64)
ap.add_argument("--max-slices", type=int, default=8)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--graph", type=int, default=-1, help="-1: contract_sliced's default mode (staged reuse); 0 / 1: slice by slice, eager / hipGraph")
a = ap.parse_args()
be = ta.get_hip_backend()
import networkx as nx
g = nx.random_regular_graph(3, a.n, seed=6)
D = a.D
nodes = {}
for v in sorted(g.nodes):
  if a.dtype == "bf16":
    t = be.device_random((D, D, D), dtype=ta.bfloat16, seed=100 + v, normal=True, a=0.0, b=D ** -1.5)
  else:
    t = be.device_random((D, D, D), dtype=np.float32, seed=100 + v, normal=True, a=0.0, b=D ** -1.5)
  nodes[v] = network.Node(t, backend=be)
slot = {v: 0 for v in g.nodes}
for x, y in sorted(g.edges):
  network.connect(nodes[x][slot[x]], nodes[y][slot[y]])
  slot[x] += 1
  slot[y] += 1
nodes = [nodes[v] for v in sorted(g.nodes)]
t0 = time.perf_counter()
cuts = distributed.choose_cut_edges(nodes, min_slices=a.min_slices)
rep = distributed.slicing_report(nodes, cuts)
t_plan = time.perf_counter() - t0


class Sub(distributed.LocalComm):
  """Pretend to be rank 0 of `world` so only every world-th slice is contracted."""
  def __init__(self, world):
    self.rank, self.world = 0, world


n_slices = int(rep["n_slices"])
world = max(1, n_slices // a.max_slices)
ug = None if a.graph < 0 else bool(a.graph)
out = distributed.contract_sliced(nodes, cuts, comm=Sub(world), use_graph=ug)  # warm-up
be.synchronize()
t0 = time.perf_counter()
stats = {}
out = distributed.contract_sliced(nodes, cuts, comm=Sub(world), use_graph=ug, stats=stats)
be.synchronize()
dt = time.perf_counter() - t0
done = len(range(0, n_slices, world))
print(json.dumps({"D": D, "dtype": a.dtype, "graph": a.graph, "n_slices": n_slices, "slices_run": done, "plan_s": t_plan,
                  "sec_per_slice": dt / done, "flops_per_slice": 2.0 * rep["flops_per_slice"],
                  "mode": stats.get("mode"), "stage_runs": stats.get("stage_runs"),
                  # executed flops (a step runs once per value of the cuts it depends on in the default mode)
                  "tflops": 2.0 * stats.get("executed_macs", rep["flops_per_slice"] * done) / dt / 1e12,
                  "tflops_if_every_slice_ran_alone": 2.0 * rep["flops_per_slice"] * done / dt / 1e12,
                  "peak_elems": rep["peak_per_slice"],
                  "est_full_1gpu_s": dt / done * n_slices, "partial": float(np.asarray(out).reshape(-1)[0])}))
