"""Every GEMM / permute / gather launch of a workload with its shape and its own duration (HIP events around each
launch, so the run is serialised: use it for WHERE the time goes, not for totals).  Aggregated by (kind, shape).
  python tools/launch_shapes.py --workload rr --D 16            (the 64-node 3-regular network, sliced)
  python tools/launch_shapes.py --workload mera --chi 32"""
import argparse, collections, ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib, distributed, network, contractors, workloads as wl

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rr")
ap.add_argument("--D", type=int, default=16)
ap.add_argument("--chi", type=int, default=32)
ap.add_argument("--min-slices", type=int, default=64)
ap.add_argument("--top", type=int, default=30)
ap.add_argument("--world", type=int, default=1, help="rr: contract only the share of --rank out of --world ranks")
ap.add_argument("--rank", type=int, default=0)
a = ap.parse_args()
be = ta.get_hip_backend()
lib = be.lib
LOG = []


def wrap(name, describe):
  real = getattr(lib, name)

  def call(*args):
    s = _lib.Event().record()
    rc = real(*args)
    e = _lib.Event().record()
    LOG.append((name, describe(*args), s, e, lib.tnh_gemm_last_kernel().decode() if "gemm" in name else ""))
    return rc
  setattr(lib, name, call)


def val(x):
  return int(getattr(x, "value", x))


def view_desc(code, oc, m, n, k, A, va, B, vb, C, ldc):
  va, vb = va._obj, vb._obj
  form = lambda v: ("K" if v.sk0 == 1 else "k-major") + (f" rows {v.r0}x{v.sr0}+{v.sr1}" if v.sr1 else "") + f" k0 {v.k0}" + (f" sk0 {v.sk0}" if v.sk0 != 1 else "")
  return f"view M={val(m)} N={val(n)} K={val(k)} a[{form(va)}] b[{form(vb)}]", 2.0 * val(m) * val(n) * val(k), 0


def gemm_desc(code, oc, ta_, tb_, m, n, k, A, lda, B, ldb, C, ldc, batch, sa, sb, sc):
  return f"gemm dt{val(code)} t{val(ta_)}{val(tb_)} M={val(m)} N={val(n)} K={val(k)} batch={val(batch)}", 2.0 * val(m) * val(n) * val(k) * val(batch), 0


def permute_desc(out, src, nd, shape, perm, item):
  nd = val(nd)
  sh = [shape[i] for i in range(nd)]
  return f"permute {sh} {[perm[i] for i in range(nd)]}", 0, 2 * int(np.prod(sh)) * val(item)


def gather_desc(code, ms, k, nl, S, lds, L, lelems, desc, C, ldc, sf):
  return f"gather Ms={val(ms)} K={val(k)} Nl={val(nl)} small_first={val(sf)}", 2.0 * val(ms) * val(k) * val(nl), 0


wrap("tnh_gemm_view", view_desc)
wrap("tnh_gemm", gemm_desc)
wrap("tnh_permute", permute_desc)
wrap("tnh_gemm_gather", gather_desc)

if a.workload == "rr":
  import networkx as nx
  g = nx.random_regular_graph(3, 64, seed=6)
  D = a.D
  nodes = {v: network.Node(be.device_random((D, D, D), dtype=ta.bfloat16, seed=100 + v, normal=True, a=0.0, b=D ** -1.5), backend=be)
           for v in sorted(g.nodes)}
  slot = {v: 0 for v in g.nodes}
  for x, y in sorted(g.edges):
    network.connect(nodes[x][slot[x]], nodes[y][slot[y]])
    slot[x] += 1
    slot[y] += 1
  nodes = [nodes[v] for v in sorted(g.nodes)]
  class Share(distributed.LocalComm):
    rank, world = a.rank, a.world
  cuts = distributed.choose_cut_edges(nodes, min_slices=a.min_slices, **({"world": a.world} if a.world > 1 else {}))
  run = lambda: distributed.contract_sliced(nodes, cuts, comm=Share())
elif a.workload == "mera64":      # 16 slices of the bond-sliced chi = 64 layer (left placement), partials reused
  layer = wl.MeraSlicedLayer(be, 64, "left", ta.bfloat16, seed=40)
  sl = layer.all_slices()[:a.D]
  run = lambda: layer.contract(sl, reuse=True)
else:
  chi = a.chi
  sc = lambda n: float(n) ** -0.5
  ham = be.device_random((chi,) * 6, dtype=ta.bfloat16, seed=1, normal=True, b=sc(chi**3))
  rho = be.device_random((chi,) * 6, dtype=ta.bfloat16, seed=2, normal=True, b=sc(chi**3))
  iso = be.device_random((chi,) * 3, dtype=ta.bfloat16, seed=3, normal=True, b=sc(chi))
  dis = be.device_random((chi,) * 4, dtype=ta.bfloat16, seed=4, normal=True, b=sc(chi * chi))
  run = lambda: wl.mera_energy(be, ham, rho, iso, dis, lambda nd: contractors.branch(nd, nbranch=2))

run()
be.synchronize()
LOG.clear()
t0 = time.perf_counter()
run()
be.synchronize()
wall = time.perf_counter() - t0
agg = collections.OrderedDict()
for name, (desc, flop, nbytes), s, e, kernel in LOG:
  ms = s.elapsed_ms(e)
  r = agg.setdefault((desc, kernel), [0, 0.0, 0.0, 0.0])
  r[0] += 1
  r[1] += ms
  r[2] += flop
  r[3] += nbytes
total = sum(r[1] for r in agg.values())
print(f"# {len(LOG)} launches, {total:.1f} ms inside the event pairs, {wall * 1e3:.1f} ms wall (serialised run)")
print(f"{'calls':>6} {'ms':>9} {'pct':>6} {'TF|TB/s':>8}  launch")
for (desc, kernel), (calls, ms, flop, nbytes) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
  rate = flop / ms / 1e9 if flop else nbytes / ms / 1e9
  print(f"{calls:6d} {ms:9.2f} {100 * ms / total:6.1f} {rate:8.1f}  {desc}  {kernel}")
