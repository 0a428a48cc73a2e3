"""Which K1 permutes does a network contraction issue, and how fast are they?
Wraps HipBackend.transpose with HIP-event timing while one slice of the rr64 network runs.
  python tools/permute_trace.py --D 16"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib, distributed, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--D", type=int, default=16)
ap.add_argument("--top", type=int, default=25)
a = ap.parse_args()
be = ta.get_hip_backend()
tensors = workloads.random_regular_device_tensors(be, 64, a.D, ta.bfloat16, seed=6)
nodes = workloads.random_regular_network(be, n=64, D=a.D, seed=6, tensors=tensors)
cuts = distributed.choose_cut_edges(nodes, min_slices=64)
rep = distributed.slicing_report(nodes, cuts)

class Sub(distributed.LocalComm):
  rank, world = 0, int(rep["n_slices"])

distributed.contract_sliced(nodes, cuts, comm=Sub(), use_graph=False)  # warm-up
be.synchronize()
log = []
orig = be.transpose
def traced(tensor, perm=None):
  s = _lib.Event().record()
  out = orig(tensor, perm)
  e = _lib.Event().record()
  log.append((s, e, tuple(tensor.shape), tuple(perm) if perm is not None else None, tensor.itemsize, tensor.size))
  return out
be.transpose = traced
t0 = time.perf_counter()
distributed.contract_sliced(nodes, cuts, comm=Sub(), use_graph=False)
be.synchronize()
dt = time.perf_counter() - t0
be.transpose = orig
rows = []
for s, e, shape, perm, isz, size in log:
  ms = s.elapsed_ms(e)
  rows.append({"ms": ms, "GBps": 2 * size * isz / ms / 1e6, "shape": shape, "perm": perm})
tot = sum(r["ms"] for r in rows)
print(json.dumps({"slice_s": dt, "n_permutes": len(rows), "permute_ms": tot,
                  "bytes_GB": sum(2 * l[5] * l[4] for l in log) / 1e9}))
for r in sorted(rows, key=lambda r: -r["ms"])[:a.top]:
  print(json.dumps(r))
