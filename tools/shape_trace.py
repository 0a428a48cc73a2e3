"""Host-only dry run of the hip backend's lowering: records every GEMM / permute the
tensordot lowering would launch for a workload (no GPU, no data).  Used to size kernels.
  python tools/shape_trace.py --D 12"""
import argparse, collections, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib, hip_backend, device_tensor, distributed, network, contractors, pathfinder

LOG = []

class FakeTensor(device_tensor.DeviceTensor):
  def __init__(self, shape, code, alias=None):
    self._shape = tuple(int(s) for s in shape); self._code = code; self._block = None; self._offset = 0; self._alias = None; self._pad = None
  @classmethod
  def empty(cls, shape, code, alias=None):
    return cls(shape, code)
  def _clone(self, shape, pad=None):
    t = FakeTensor(shape, self._code)
    t._pad = pad
    return t
  @property
  def ptr(self):
    return 0

device_tensor.DeviceTensor.empty = FakeTensor.empty
hip_backend.DeviceTensor.empty = FakeTensor.empty
device_tensor.DeviceTensor._fresh = classmethod(lambda cls, shape, code, nbytes, alias=None: FakeTensor(shape, code))

class FakeLib:
  """only what the gather lowering calls"""
  def tnh_gemm_gather(self, code, ms, k, nl, s, lds, l, l_elems, desc, c, ldc, small_first):
    d = desc._obj
    LOG.append(("gather_gemm", int(ms), int(nl), int(k), hip_backend._gather_piece_bytes(d), bool(d.k_mask & 1),
                int(np.prod([d.ext[i] for i in range(d.nd) if not (d.k_mask >> i) & 1]))))
    return _lib.OK

class TraceBackend(hip_backend.HipBackend):
  @property
  def lib(self):
    return FakeLib()
  def cast(self, tensor, dtype):
    return tensor
  def transpose(self, tensor, perm=None):
    nd = tensor.ndim
    perm = tuple(range(nd - 1, -1, -1)) if perm is None else tuple(perm)
    if perm == tuple(range(nd)):
      return tensor
    LOG.append(("permute", tensor.shape, perm))
    return FakeTensor([tensor.shape[p] for p in perm], tensor.code)
  def _tensordot_in_place(self, *args, **kwargs):
    return None   # the dry run records the classic lowering (the view GEMM needs >= 192 tiles; D = 12 never has them)
  def _gemm(self, a, b, trans_a, trans_b, m, n, k, lda, ldb, batch=1, stride_a=0, stride_b=0, out_shape=None, out_code=None,
            alias=None):
    LOG.append(("gemm", int(trans_a), int(trans_b), m, n, k, batch))
    return FakeTensor(out_shape if out_shape is not None else (m, n), a.code)
  def _outer(self, a, b, out_shape, alias=None):
    LOG.append(("outer", a.size, b.size))
    return FakeTensor(out_shape, a.code)
  def _strided_copy(self, tensor, shape, strides, offset):
    LOG.append(("gather", tuple(shape)))
    return FakeTensor(shape, tensor.code)
  def addition(self, a, b):
    return a
  def multiply(self, a, b):
    return a

ap = argparse.ArgumentParser()
ap.add_argument("--D", type=int, default=12)
ap.add_argument("--min-slices", type=int, default=64)
ap.add_argument("--gather", type=int, default=1, help="0: without the gather lowering (tnh_gemm_gather)")
a = ap.parse_args()
be = TraceBackend()
be.gather_gemm = bool(a.gather)
import networkx as nx
g = nx.random_regular_graph(3, 64, seed=6)
D = a.D
nodes = {v: network.Node(FakeTensor((D, D, D), _lib.BF16), backend=be) for v in sorted(g.nodes)}
slot = {v: 0 for v in g.nodes}
for x, y in sorted(g.edges):
  network.connect(nodes[x][slot[x]], nodes[y][slot[y]]); slot[x] += 1; slot[y] += 1
nodes = [nodes[v] for v in sorted(g.nodes)]
cuts = distributed.choose_cut_edges(nodes, min_slices=a.min_slices)
class One(distributed.LocalComm):
  rank, world = 0, 10**9
STATS = {}
distributed.contract_sliced(nodes, cuts, comm=One(), use_graph=False, stats=STATS)
rep = distributed.slicing_report(nodes, cuts)
print("slice-invariant steps (run once, before the slice loop):", STATS.get("hoisted_steps"), "of", rep["steps_per_slice"],
      f"= {rep['flops_invariant_per_slice'] / rep['flops_per_slice']:.2e} of a slice's flops; the launches below are those steps + ONE slice")
tot = 0
for rec in LOG:
  if rec[0] == "gemm":
    _, ta_, tb_, m, n, k, batch = rec
    fl = 2.0 * m * n * k * batch; tot += fl
  elif rec[0] == "gather_gemm":
    tot += 2.0 * rec[1] * rec[2] * rec[3]
for rec in LOG:
  if rec[0] == "gemm":
    _, ta_, tb_, m, n, k, batch = rec
    fl = 2.0 * m * n * k * batch
    if fl / tot > 0.002:
      print(f"gemm tA={ta_} tB={tb_} M={m} N={n} K={k}  {fl:.3e} flop  {100*fl/tot:.1f}%")
  elif rec[0] == "gather_gemm":
    fl = 2.0 * rec[1] * rec[2] * rec[3]
    if fl / tot > 0.002:
      print(f"gather_gemm Ms={rec[1]} Nl={rec[2]} K={rec[3]}  {fl:.3e} flop  {100*fl/tot:.1f}%  pieces of {rec[4]} B, "
            f"innermost axis {'contracted' if rec[5] else 'free'}, BN {rec[6]}")
  elif rec[0] == "permute" and np.prod(rec[1]) > 1e6:
    print("permute", rec[1], rec[2], f"{np.prod(rec[1]):.2e} elems")
print("total flop", f"{tot:.3e}", "n_gemm", sum(1 for r in LOG if r[0] == "gemm"), "n_gather", sum(1 for r in LOG if r[0] == "gather_gemm"),
      "permuted elements", f"{sum(float(np.prod(r[1])) for r in LOG if r[0] == 'permute'):.3e}")
