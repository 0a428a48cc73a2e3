"""configs[4] at chi = 64: the dense network does not fit (two rank-6 inputs of 137 GB bf16 each, chi^7
intermediates), so the layer energy is bond-sliced: one cut on a leg of the hamiltonian and one on a leg
of the state give chi^2 = 4096 independent slices per placement whose largest tensor is chi^5.  This
probe measures the PER-SLICE cost on one GPU: it rebuilds the 12-node topology with the two cut bonds at
dimension 1 (operands generated directly at their sliced shapes: synthetic data, no weight sharing),
contracts it with the branch(nbranch=2) path of the sliced sizes, and extrapolates to the whole layer
on N GPUs (slices are independent; one all-reduce of a scalar at the end).
  python tools/mera_slice_sample.py --chi 64 [--reps 3] [--gpus 8]"""
import argparse, functools, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import contractors, network, pathfinder, workloads as wl

ap = argparse.ArgumentParser()
ap.add_argument("--chi", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--gpus", type=int, default=8)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
be = ta.get_hip_backend()
dt = ta.bfloat16 if a.dtype == "bf16" else np.float32
chi = a.chi


class Shape:   # shape-only stand-in for planning (never touched by a kernel)
  def __init__(self, shape): self.shape = tuple(shape); self.dtype = np.dtype(np.float32); self.ndim = len(shape)


class PlanBackend:
  name = "plan"
  def convert_to_tensor(self, t): return t
  def shape_tuple(self, t): return t.shape
  def conj(self, t): return t


algo = functools.partial(pathfinder.branch, nbranch=2)
out = {"chi": chi, "dtype": a.dtype, "placements": {}}
total_s = 0.0
for placement in ("left", "right"):
  pb = PlanBackend()
  ham, rho = Shape((chi,) * 6), Shape((chi,) * 6)
  plan_nodes = wl.mera_layer_network(pb, ham, rho, Shape((chi,) * 3), Shape((chi,) * 4), placement)
  inputs = [set(n.edges) for n in plan_nodes]
  sizes = {e: e.dimension for e in network.get_all_edges(plan_nodes)}
  hnode = [n for n in plan_nodes if n.tensor is ham][0]
  rnode = [n for n in plan_nodes if n.tensor is rho][0]
  best = None
  for eh in hnode.edges:                      # cheapest pair of cuts: one leg of h, one leg of rho
    for er in rnode.edges:
      trial = dict(sizes); trial[eh] = 1; trial[er] = 1
      path = algo(inputs, set(), trial)
      flops, peak = pathfinder.path_cost(inputs, set(), trial, path)
      if best is None or (flops, peak) < best[0]:
        best = ((flops, peak), eh, er, path)
  (flops, peak), eh, er, path = best
  cut = {id(eh), id(er)}
  # real nodes at the sliced shapes, same topology
  index = {id(n): i for i, n in enumerate(plan_nodes)}
  real = []
  for i, n in enumerate(plan_nodes):
    shape = tuple(1 if id(e) in cut else e.dimension for e in n.edges)
    scale = float(np.prod(shape)) ** -0.25
    real.append(network.Node(be.device_random(shape, dtype=dt, seed=17 * i + 3, normal=True, a=0.0, b=scale), backend=be))
  done = set()
  for n in plan_nodes:
    for ax, e in enumerate(n.edges):
      if id(e) in done or e.is_dangling():
        continue
      done.add(id(e))
      (n1, a1), (n2, a2) = e.ends()
      network.connect(real[index[id(n1)]][a1], real[index[id(n2)]][a2])
  best_t = None
  for _ in range(a.reps + 1):
    node_map, _ = network.copy(real)
    be.synchronize(); t0 = time.perf_counter()
    res = contractors.contract_path(path, [node_map[n] for n in real]).tensor
    be.synchronize(); t = time.perf_counter() - t0
    best_t = t if best_t is None else min(best_t, t)
    import ctypes
    iu, ca, pk = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    be.lib.tnh_mem_stats(ctypes.byref(iu), ctypes.byref(ca), ctypes.byref(pk))
    print(f"# {placement} rep {t:.4f} s  in_use {iu.value/2**30:.1f} GiB cached {ca.value/2**30:.1f} GiB peak {pk.value/2**30:.1f} GiB", file=sys.stderr, flush=True)
  n_slices = chi * chi
  out["placements"][placement] = {"n_slices": n_slices, "macs_per_slice": float(flops), "peak_elems_per_slice": float(peak),
                                  "sec_per_slice": best_t, "tflops": 2.0 * float(flops) / best_t / 1e12}
  total_s += best_t * n_slices
out["est_layer_energy_1gpu_s"] = total_s
out[f"est_layer_energy_{a.gpus}gpu_s"] = total_s / a.gpus
print(json.dumps(out))
