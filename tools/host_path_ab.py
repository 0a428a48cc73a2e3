"""Round 6 A/B on one box: bench.py's sweep step (two Nodes, two connects, contract_between) with the host fast paths
of round 6 on and off (`HipBackend.plan_cache`, `Node._from_contraction`), at sizes from host-bound (D = 32) to
kernel-bound (D = 128 L1, D = 192 L0).  Alternating order, several rounds: box drift shows as spread between rounds.
  python tools/host_path_ab.py [--rows 32:L0,32:L1,64:L0,128:L1,192:L0] [--rounds 3]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import tensornetwork_amd as ta
from tensornetwork_amd import network, device_tensor

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="32:L0,32:L1,64:L0,128:L1,192:L0")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
be = ta.get_hip_backend()
fast_ctor = network.Node._from_contraction


def slow_ctor(tensor, name, backend, sources):
  out = network.Node(tensor, name=name, backend=backend)
  network._adopt_edges(out, sources)
  return out


for row in a.rows.split(","):
  D, layout = row.split(":")
  D = int(D)
  A, B = bench.make_nodes(ta, be, D, "L0", seed=7, fill="normal")
  reps = 3 if D >= 192 else 10
  for rnd in range(a.rounds):
    for mode in ("fast", "slow"):
      be.plan_cache = mode == "fast"
      network.Node._from_contraction = fast_ctor if mode == "fast" else staticmethod(slow_ctor)
      g0 = device_tensor.gc_stats()
      t, permutes = bench.timed_steps(be, lambda: bench.one_step(ta, be, A, B, layout), reps, batches=1 if D >= 192 else 3)
      g1 = device_tensor.gc_stats()
      print(json.dumps({"D": D, "layout": layout, "round": rnd, "mode": mode, "ms": round(t * 1e3, 4),
                        "tflops": round(2.0 * D**6 / t / 1e12, 1), "permutes": permutes,
                        "gc_full_passes": g1["full_passes"] - g0["full_passes"]}), flush=True)
  be.plan_cache = True
  network.Node._from_contraction = fast_ctor
  del A, B
