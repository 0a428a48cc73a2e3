"""Config 5: binary-MERA layer energy (12-node network, contractors.branch nbranch=2) on one GPU.
  python tools/mera_probe.py --chi 4,8,16 [--dtype bf16|f32]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import contractors, pathfinder, network, workloads as wl

ap = argparse.ArgumentParser()
ap.add_argument("--chi", default="4,8,16")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--slices", type=int, default=0, help="bond-slice each placement into >= this many slices (0 = unsliced)")
ap.add_argument("--max-slices", type=int, default=0, help="with --slices: run only this many slices per placement (timing sample)")
a = ap.parse_args()
be = ta.get_hip_backend()
dt = ta.bfloat16 if a.dtype == "bf16" else np.float32
for chi in [int(c) for c in a.chi.split(",")]:
  # operands in HBM: random stand-ins of the right shape (isometry constraints do not change the cost)
  sc = lambda n: float(n) ** -0.5
  ham = be.device_random((chi,) * 6, dtype=dt, seed=1, normal=True, b=sc(chi**3))
  rho = be.device_random((chi,) * 6, dtype=dt, seed=2, normal=True, b=sc(chi**3))
  iso = be.device_random((chi,) * 3, dtype=dt, seed=3, normal=True, b=sc(chi))
  dis = be.device_random((chi,) * 4, dtype=dt, seed=4, normal=True, b=sc(chi * chi))
  nodes = wl.mera_layer_network(be, ham, rho, iso, dis, "left")
  inputs = [set(n.edges) for n in nodes]
  sizes = {e: e.dimension for e in network.get_all_edges(nodes)}
  path = pathfinder.branch(inputs, set(), sizes, nbranch=2)
  flops, peak = pathfinder.path_cost(inputs, set(), sizes, path)
  best = None
  extra = {}
  if a.slices > 0:
    import functools
    from tensornetwork_amd import distributed
    algo = functools.partial(pathfinder.branch, nbranch=2)
    cuts = distributed.choose_cut_edges(nodes, min_slices=a.slices, algorithm=algo)
    rep = distributed.slicing_report(nodes, cuts, algorithm=algo)
    n_slices = int(rep["n_slices"])
    world = max(1, n_slices // a.max_slices) if a.max_slices else 1
    class Sub(distributed.LocalComm):
      rank = 0
    Sub.world = world
    done = len(range(0, n_slices, world))
    extra = {"n_slices": n_slices, "slices_run": done, "cut_bonds": len(cuts), "peak_per_slice": rep["peak_per_slice"],
             "flops_per_slice": rep["flops_per_slice"], "slicing_overhead": rep["overhead"]}
    def run():
      outs = []
      for pl in ("left", "right"):
        nd = wl.mera_layer_network(be, ham, rho, iso, dis, pl)
        c = distributed.choose_cut_edges(nd, min_slices=a.slices, algorithm=algo)
        outs.append(distributed.contract_sliced(nd, c, comm=Sub(), algorithm=algo))
      return be.multiply(be.addition(outs[0], outs[1]), 0.5)
    flops = rep["flops_per_slice"] * done
  else:
    run = lambda: wl.mera_energy(be, ham, rho, iso, dis, lambda nd: contractors.branch(nd, nbranch=2))
  for _ in range(a.reps + 1):
    be.synchronize()
    t0 = time.perf_counter()
    e = run()
    be.synchronize()
    t = time.perf_counter() - t0
    best = t if best is None else min(best, t)
  # path_cost counts multiply-adds per placement: x2 flop, x2 placements (sliced mode: `flops` already sums the slices run)
  mult = 4.0 if a.slices == 0 else 2.0 * 2.0
  rec = {"chi": chi, "dtype": a.dtype, "sec": best, "flops_2placements": mult * float(flops),
         "tflops": mult * float(flops) / best / 1e12, "peak_elems": float(peak),
         "energy": float(np.asarray(e).reshape(-1)[0])}
  rec.update(extra)
  print(json.dumps(rec), flush=True)
  del ham, rho, nodes, e
