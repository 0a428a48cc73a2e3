"""The BASELINE.json workloads as network builders (shared by bench.py, tools/ and tests/).

Every builder takes a backend and returns freshly connected ``network.Node`` lists,
so the same topology can be contracted by the hip backend and by the CPU oracle.

  random_regular_network  north-star scaling network: one rank-3 tensor per vertex of a
                          random 3-regular graph, closed -> scalar.
  mps_overlap_network     BASELINE configs[3]: <psi|psi> of an open-boundary MPS
                          (kets + conjugated bras), `contractors.greedy` target.
  mera_layer_network      BASELINE configs[4]: the 12-node binary-MERA layer energy network
                          of examples/simple_mera/simple_mera.py:54-112 ('left'/'right').
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np

from tensornetwork_amd import network


# ------------------------------------------------------------------ 3-regular graph
def regular_graph_edges(n: int, degree: int = 3, seed: int = 6) -> List[Tuple[int, int]]:
  """Edge list of a random `degree`-regular simple graph on n vertices, sorted.

  networkx.random_regular_graph(degree, n, seed) when networkx is importable (the graph
  SURVEY.md 8d names); otherwise a seeded pairing model with rejection (same properties,
  different instance)."""
  try:
    import networkx as nx  # pylint: disable=import-outside-toplevel
    return sorted(tuple(sorted(e)) for e in nx.random_regular_graph(degree, n, seed=seed).edges)
  except ImportError:
    pass
  rng = np.random.default_rng(seed)
  while True:
    stubs = np.repeat(np.arange(n), degree)
    rng.shuffle(stubs)
    pairs = stubs.reshape(-1, 2)
    edges = {tuple(sorted(p)) for p in pairs.tolist()}
    if len(edges) == len(pairs) and all(a != b for a, b in edges):
      return sorted(edges)


def random_regular_network(be, n: int = 64, D: int = 4, seed: int = 6, dtype=np.float32,
                           tensors: Sequence = None, scale: float = None):
  """Closed network on a random 3-regular graph; entries N(0, scale^2), scale = D^(-3/4)
  by default so that the contracted scalar is O(1) (variance D^(3n/2) scale^(2n))."""
  edges = regular_graph_edges(n, 3, seed)
  if tensors is None:
    rng = np.random.default_rng(seed)
    sc = float(D) ** -0.75 if scale is None else scale
    tensors = [(rng.standard_normal((D, D, D)) * sc).astype(dtype) for _ in range(n)]
  nodes = [network.Node(t, backend=be) for t in tensors]
  slot = [0] * n
  for x, y in edges:
    network.connect(nodes[x][slot[x]], nodes[y][slot[y]])
    slot[x] += 1
    slot[y] += 1
  return nodes


def random_regular_device_tensors(be, n: int, D: int, dtype, seed: int = 6, scale: float = None):
  """Operands of `random_regular_network` generated in HBM (bench sizes)."""
  sc = float(D) ** -0.75 if scale is None else scale
  return [be.device_random((D, D, D), dtype=dtype, seed=1000 * seed + v, normal=True, a=0.0, b=sc)
          for v in range(n)]


# ------------------------------------------------------------------------ MPS chain
def mps_tensors(n_sites: int, d: int, D: int, seed: int = 5, dtype=np.float32):
  rng = np.random.default_rng(seed)
  dims = [1] + [D] * (n_sites - 1) + [1]
  return [(rng.standard_normal((dims[i], d, dims[i + 1])) / np.sqrt(d * max(dims[i], 1))).astype(dtype)
          for i in range(n_sites)]


def mps_overlap_network(be, kets: Sequence):
  """2 n nodes: kets A_i[l, p, r] and their conjugates, physical and bond legs connected."""
  n = len(kets)
  nk = [network.Node(k, backend=be) for k in kets]
  nb = [network.Node(np.conj(k) if isinstance(k, np.ndarray) else be.conj(k), backend=be) for k in kets]
  for i in range(n):
    network.connect(nk[i][1], nb[i][1])
    if i + 1 < n:
      network.connect(nk[i][2], nk[i + 1][0])
      network.connect(nb[i][2], nb[i + 1][0])
  network.connect(nk[0][0], nb[0][0])
  network.connect(nk[-1][2], nb[-1][2])
  return nk + nb


# ----------------------------------------------------------------------- MERA layer
# (node, axis) -- (node, axis) tables restating the wiring of
# examples/simple_mera/simple_mera.py:71-107.  Node names: w* isometries (rank 3:
# two lower legs, one upper), u* disentanglers (rank 4), 'c' suffix = conjugate copy,
# h = 3-site hamiltonian (rank 6), rho = 3-site reduced state (rank 6).
_MERA_COMMON = [
    ("wl", 2, "rho", 0), ("wc", 2, "rho", 1), ("wr", 2, "rho", 2),
    ("wl", 0, "wlc", 0), ("wl", 1, "ul", 2), ("wc", 0, "ul", 3),
    ("wc", 1, "ur", 2), ("wr", 0, "ur", 3), ("wr", 1, "wrc", 1),
    ("ulc", 2, "wlc", 1), ("ulc", 3, "wcc", 0), ("urc", 2, "wcc", 1), ("urc", 3, "wrc", 0),
    ("wlc", 2, "rho", 3), ("wcc", 2, "rho", 4), ("wrc", 2, "rho", 5),
]
_MERA_PLACEMENT = {
    "right": [("ul", 0, "ulc", 0), ("ul", 1, "h", 3), ("ur", 0, "h", 4), ("ur", 1, "h", 5),
              ("h", 0, "ulc", 1), ("h", 1, "urc", 0), ("h", 2, "urc", 1)],
    "left": [("ul", 0, "h", 3), ("ul", 1, "h", 4), ("ur", 0, "h", 5), ("ur", 1, "urc", 1),
             ("h", 0, "ulc", 0), ("h", 1, "ulc", 1), ("h", 2, "urc", 0)],
}


def mera_layer_network(be, hamiltonian, state, isometry, disentangler, placement: str):
  """The 12 connected nodes of one binary-MERA layer energy term (closed network).

  With ``hamiltonian=None`` the h node is left out and its six partner legs stay
  dangling: contracting that network gives dE/dh, i.e. the descended state of
  simple_mera.py:117-128 (the reference gets it from jax.grad); the second return value
  is then the dangling edges in the order of h's axes."""
  conj = lambda t: np.conj(t) if isinstance(t, np.ndarray) else be.conj(t)
  mk = lambda t: network.Node(t, backend=be)
  nodes: Dict[str, network.Node] = {
      "wl": mk(isometry), "wc": mk(isometry), "wr": mk(isometry),
      "wlc": mk(conj(isometry)), "wcc": mk(conj(isometry)), "wrc": mk(conj(isometry)),
      "ul": mk(disentangler), "ur": mk(disentangler),
      "ulc": mk(conj(disentangler)), "urc": mk(conj(disentangler)),
      "rho": mk(state),
  }
  if hamiltonian is not None:
    nodes["h"] = mk(hamiltonian)
  open_legs = {}
  for a, i, b, j in _MERA_COMMON + _MERA_PLACEMENT[placement]:
    if hamiltonian is None and "h" in (a, b):
      other, oax, hax = (b, j, i) if a == "h" else (a, i, j)
      open_legs[hax] = nodes[other][oax]
      continue
    network.connect(nodes[a][i], nodes[b][j])
  if hamiltonian is None:
    return list(nodes.values()), [open_legs[ax] for ax in range(6)]
  return list(nodes.values())


def mera_energy(be, hamiltonian, state, isometry, disentangler, contractor):
  """0.5 * (left + right) layer energy, simple_mera.py:53-112; `contractor(nodes)` -> Node."""
  out = [contractor(mera_layer_network(be, hamiltonian, state, isometry, disentangler, pl)).tensor
         for pl in ("left", "right")]
  return be.multiply(be.addition(out[0], out[1]), 0.5)


def mera_descend(be, state, isometry, disentangler, contractor):
  """Descending super-operator: average over placements of the layer network with h removed.
  `contractor(nodes, output_edge_order)` -> Node."""
  out = []
  for pl in ("left", "right"):
    nodes, legs = mera_layer_network(be, None, state, isometry, disentangler, pl)
    out.append(contractor(nodes, legs).tensor)
  return be.multiply(be.addition(out[0], out[1]), 0.5)


class _ShapeOnly:
  """shape-only stand-in used for planning (never touched by a kernel)."""

  def __init__(self, shape):
    self.shape = tuple(shape)
    self.dtype = np.dtype(np.float32)
    self.ndim = len(self.shape)


class _PlanBackend:
  name = "plan"

  def convert_to_tensor(self, t):
    return t

  def shape_tuple(self, t):
    return t.shape

  def conj(self, t):
    return t


def _mera_slice_plan(chi: int, placement: str):
  """Host-only plan of ONE bond-sliced placement of the MERA layer at bond dimension chi: the 12-node topology with
  shape-only tensors, the cheapest pair of cuts (one leg of the hamiltonian, one leg of the state) under
  branch(nbranch=2), the path on the sliced sizes and its cost."""
  import functools  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import pathfinder  # pylint: disable=import-outside-toplevel
  algo = functools.partial(pathfinder.branch, nbranch=2)
  pb = _PlanBackend()
  ham, rho = _ShapeOnly((chi,) * 6), _ShapeOnly((chi,) * 6)
  plan_nodes = mera_layer_network(pb, ham, rho, _ShapeOnly((chi,) * 3), _ShapeOnly((chi,) * 4), placement)
  inputs = [set(n.edges) for n in plan_nodes]
  sizes = {e: e.dimension for e in network.get_all_edges(plan_nodes)}
  hnode = [n for n in plan_nodes if n.tensor is ham][0]
  rnode = [n for n in plan_nodes if n.tensor is rho][0]
  best = None
  for eh in hnode.edges:                      # cheapest pair of cuts: one leg of h, one leg of rho
    for er in rnode.edges:
      trial = dict(sizes)
      trial[eh] = 1
      trial[er] = 1
      path = algo(inputs, set(), trial)
      flops, peak = pathfinder.path_cost(inputs, set(), trial, path)
      if best is None or (flops, peak) < best[0]:
        best = ((flops, peak), eh, er, path)
  (flops, peak), eh, er, path = best
  return {"nodes": plan_nodes, "hnode": hnode, "rnode": rnode, "cut": {id(eh), id(er)}, "path": path,
          "flops": float(flops), "peak": float(peak), "depth": pathfinder.path_depth(path, len(plan_nodes))}


def _mera_slice_network(be, plan, tensor_of):
  """Real nodes of one slice: node i of the plan gets tensor_of(i, node, sliced_shape)."""
  plan_nodes, cut = plan["nodes"], plan["cut"]
  index = {id(n): i for i, n in enumerate(plan_nodes)}
  real = []
  for i, n in enumerate(plan_nodes):
    shape = tuple(1 if id(e) in cut else e.dimension for e in n.edges)
    real.append(network.Node(tensor_of(i, n, shape), backend=be))
  done = set()
  for n in plan_nodes:
    for e in n.edges:
      if id(e) in done or e.is_dangling():
        continue
      done.add(id(e))
      (n1, a1), (n2, a2) = e.ends()
      network.connect(real[index[id(n1)]][a1], real[index[id(n2)]][a2])
  return real


def mera_sliced_sample(be, chi: int, dtype, reps: int = 2, seed: int = 17):
  """configs[4] at a bond dimension whose dense network does not fit (chi = 64: rank-6 inputs of 137 GB,
  chi^7 intermediates): the layer energy is bond-sliced -- one cut on a leg of the hamiltonian and one on a
  leg of the state give chi^2 independent slices per placement whose largest tensor is chi^5.  Measures the
  PER-SLICE cost on this GPU: the 12-node topology is rebuilt with the two cut bonds at dimension 1
  (operands generated directly at their sliced shapes: synthetic data), contracted with the
  branch(nbranch=2) path of the sliced sizes.  Returns per-placement slice counts, multiply-adds and
  seconds per slice; the whole layer is n_slices x that (slices are independent, one scalar all-reduce)."""
  import time  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import contractors  # pylint: disable=import-outside-toplevel
  out = {}
  for placement in ("left", "right"):
    plan = _mera_slice_plan(chi, placement)

    def tensor_of(i, node, shape):
      scale = float(np.prod(shape)) ** -0.25
      return be.device_random(shape, dtype=dtype, seed=seed * i + 3, normal=True, a=0.0, b=scale)

    real = _mera_slice_network(be, plan, tensor_of)
    best_t = None
    for _ in range(reps + 1):
      node_map, _ = network.copy(real)
      be.synchronize()
      t0 = time.perf_counter()
      res = contractors.contract_path(plan["path"], [node_map[n] for n in real]).tensor
      be.synchronize()
      t = time.perf_counter() - t0
      best_t = t if best_t is None else min(best_t, t)
      del res
    out[placement] = {"n_slices": chi * chi, "macs_per_slice": plan["flops"], "peak_elems_per_slice": plan["peak"],
                      "sec_per_slice": best_t, "tflops": 2.0 * plan["flops"] / best_t / 1e12}
    for n in real:
      n.tensor = None
      n.edges = []
  return out


def mera_slice_values(be, chi: int, slices, half_dtype, seed: int = 40):
  """Real slices of ONE chi-consistent binary-MERA layer network, each contracted three ways.

  The layer's hamiltonian and state (rank 6, 137 GB each at chi = 64) are DEFINED slice-wise along their cut legs:
  h[.., i, ..] = R(seed_h + i), rho[.., j, ..] = R(seed_r + j) with R a counter-based generator of the sliced shape
  -- so slice (i, j) of the network is exactly what `slice_edge(cut_h, i); slice_edge(cut_rho, j)` would leave of the
  full tensors, without ever materialising them; isometry and disentangler are the same full tensors in every
  slice.  For each placement and each (i, j): the slice's partial energy in `half_dtype` and in f32 on the same
  (half-rounded) values.  Returns {placement: {"depth": d, "rows": [{"slice": (i, j), "half": .., "f32": ..}]}}."""
  from tensornetwork_amd import contractors  # pylint: disable=import-outside-toplevel
  out = {}
  for placement in ("left", "right"):
    plan = _mera_slice_plan(chi, placement)
    rows = []
    for (i, j) in slices:
      def tensor_of(k, node, shape, i=i, j=j):
        scale = float(np.prod(shape)) ** -0.25
        if node is plan["hnode"]:
          sd = seed + 1000 + i
        elif node is plan["rnode"]:
          sd = seed + 5000 + j
        else:
          sd = seed + 7 * k        # isometries / disentanglers: slice-independent
        return be.device_random(shape, dtype=half_dtype, seed=sd, normal=True, a=0.0, b=scale)

      real = _mera_slice_network(be, plan, tensor_of)
      vals = {}
      for name in ("half", "f32"):
        node_map, _ = network.copy(real)
        nodes = [node_map[n] for n in real]
        if name != "half":
          for nd in nodes:
            nd.tensor = be.cast(nd.tensor, np.float32)
        res = contractors.contract_path(plan["path"], nodes).tensor
        vals[name] = float(np.asarray(res, dtype=np.float64).reshape(-1)[0])
        del res, nodes, node_map
      rows.append({"slice": [int(i), int(j)], **vals})
      for n in real:
        n.tensor = None
        n.edges = []
    out[placement] = {"depth": plan["depth"], "rows": rows}
  return out


def _mera_stages(plan):
  """Which steps of the placement's path depend on which slice index: the hamiltonian slice changes with i only, the
  state slice with j only, everything else with neither, and a step inherits the union of its operands' dependencies.
  Returns the steps as (id_a, id_b, id_new, dep) -- ids 0 .. 11 for the inputs, 12 + s for the result of step s, dep a
  string out of "", "i", "j", "ij" -- with each step's multiply-adds, the axis labels of the inputs (one label per plan
  edge) and the step at which the path contracts each label (layout planning)."""
  from tensornetwork_amd import contractors, pathfinder  # pylint: disable=import-outside-toplevel
  nodes, path, cut = plan["nodes"], plan["path"], plan["cut"]
  n = len(nodes)
  sizes = {e: (1 if id(e) in cut else e.dimension) for e in network.get_all_edges(nodes)}
  deps = [""] * n
  deps[nodes.index(plan["hnode"])] = "i"
  deps[nodes.index(plan["rnode"])] = "j"
  ids = list(range(n))
  remaining = [frozenset(nd.edges) for nd in nodes]
  dep_of = list(deps)
  steps = []
  for pair in path:
    if len(pair) == 1:
      continue
    a, b = sorted(pair)
    k1, k2 = remaining[a], remaining[b]
    others = set()
    for t, k in enumerate(remaining):
      if t not in (a, b):
        others |= k
    dep = "".join(sorted(set(dep_of[ids[a]]) | set(dep_of[ids[b]])))
    new = n + len(steps)
    dep_of.append(dep)
    steps.append((ids[a], ids[b], new, dep, float(pathfinder._size(k1 | k2, sizes))))      # pylint: disable=protected-access
    ids = [x for t, x in enumerate(ids) if t not in (a, b)] + [new]
    remaining = [k for t, k in enumerate(remaining) if t not in (a, b)] + [frozenset(d for d in (k1 | k2) if d in others)]
  times = contractors._edge_times(path, nodes)      # pylint: disable=protected-access
  return {"steps": steps, "input_dep": deps, "labels": [[id(e) for e in nd.edges] for nd in nodes],
          "shapes": [tuple(1 if id(e) in cut else e.dimension for e in nd.edges) for nd in nodes],
          "label_time": {id(e): t for e, t in times.items()},
          "macs": {d: sum(st[4] for st in steps if st[3] == d) for d in ("", "i", "j", "ij")}}


def _run_stage(be, operands, steps, label_time):
  """contractors.contract_labelled (kept under this name for the MERA stages)"""
  from tensornetwork_amd import contractors  # pylint: disable=import-outside-toplevel
  return contractors.contract_labelled(be, operands, steps, label_time)


def mera_sliced_run(be, chi: int, placement: str, half_dtype, seed: int = 40, budget_seconds: float = None,
                    check_every: int = 0, reuse_partials: bool = True):
  """ONE placement of the bond-sliced binary-MERA layer energy as a RUN: all chi^2 slices of the
  network `mera_slice_values` defines (hamiltonian and state given slice-wise along their cut legs -- what
  `slice_edge(cut_h, i); slice_edge(cut_rho, j)` leaves of the full rank-6 tensors, which at chi = 64 are 137 GB each
  and are never materialised; isometry and disentangler are the same tensors in every slice), each contracted with
  the branch(nbranch=2) path of the sliced sizes, partial energies added in f32 ON THE DEVICE (one read-back at the
  end).  The hamiltonian slice is generated once per i, the state slice once per (i, j) (2 GB of counter-based
  random numbers: ~1 ms beside a ~37 ms slice).

  budget_seconds: stop after the first slice that ends beyond it (the record then says how many slices ran).
  check_every: every that many slices the same slice is also contracted in f32 on the same half-rounded values
  (returned as `checks`: [(i, j), half, f32]) -- a sample of the a-priori rounding check, not part of the timing.
  Returns {"n_slices", "slices_done", "seconds", "energy_partial_sum", "macs_per_slice", "checks", ...}.

  reuse_partials (default): the hamiltonian slice depends on i only and the state slice on j only, so most steps of
  the path depend on ONE of the two indices (chi = 64: 40 % of a slice's multiply-adds on i, 59.6 % on j, 0.3 % on
  both, `_mera_stages`).  Those partial contractions are computed once per i (kept: chi tensors of chi^4 elements)
  and once per j (one chi^5 tensor at a time), and a slice (i, j) only runs the steps that depend on both -- the same
  steps on the same values in the same order as the slice-by-slice run, so every slice result is the same tensor.
  `executed_macs` counts what ran; `macs_per_slice` stays the cost of one slice contracted on its own."""
  import time  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import contractors  # pylint: disable=import-outside-toplevel
  plan = _mera_slice_plan(chi, placement)
  if reuse_partials:
    return _mera_sliced_run_staged(be, chi, placement, plan, half_dtype, seed, budget_seconds, check_every)
  fixed = {}

  def gen(k, node, shape, i, j):
    scale = float(np.prod(shape)) ** -0.25
    if node is plan["hnode"]:
      sd = seed + 1000 + i
    elif node is plan["rnode"]:
      sd = seed + 5000 + j
    else:
      sd = seed + 7 * k        # isometries / disentanglers: slice-independent
    return be.device_random(shape, dtype=half_dtype, seed=sd, normal=True, a=0.0, b=scale)

  acc = be.zeros((), dtype=np.float32)
  done, checks, check_seconds = 0, [], 0.0
  be.synchronize()
  t0 = time.perf_counter()
  stop = False
  for i in range(chi):
    h_i = None
    for j in range(chi):
      def tensor_of(k, node, shape, i=i, j=j):
        nonlocal h_i
        if node is plan["hnode"]:
          if h_i is None:
            h_i = gen(k, node, shape, i, j)
          return h_i
        if node is plan["rnode"]:
          return gen(k, node, shape, i, j)
        if k not in fixed:
          fixed[k] = gen(k, node, shape, i, j)
        return fixed[k]

      real = _mera_slice_network(be, plan, tensor_of)
      res = contractors.contract_path(plan["path"], real).tensor
      acc = be.addition(acc, be.reshape(be.cast(res, np.float32), ()))
      done += 1
      if check_every and (done - 1) % check_every == 0:
        be.synchronize()
        tc = time.perf_counter()
        real32 = _mera_slice_network(be, plan, lambda k, node, shape: be.cast(tensor_of(k, node, shape), np.float32))
        r32 = contractors.contract_path(plan["path"], real32).tensor
        checks.append([[i, j], float(np.asarray(res, dtype=np.float64).reshape(-1)[0]),
                       float(np.asarray(r32, dtype=np.float64).reshape(-1)[0])])
        for n in real32:
          n.tensor, n.edges = None, []
        del real32, r32
        be.synchronize()
        check_seconds += time.perf_counter() - tc
      for n in real:               # Node <-> Edge cycles: without this the 2 GB state slice waits for the cyclic GC
        n.tensor, n.edges = None, []
      del res, real
      if budget_seconds is not None and done % 16 == 0:
        be.synchronize()
        if time.perf_counter() - t0 - check_seconds > budget_seconds:
          stop = True
          break
    if stop:
      break
  be.synchronize()
  seconds = time.perf_counter() - t0 - check_seconds
  return {"placement": placement, "n_slices": chi * chi, "slices_done": done, "seconds": seconds,
          "energy_partial_sum": float(np.asarray(acc, dtype=np.float64).reshape(-1)[0]),
          "macs_per_slice": plan["flops"], "peak_elems_per_slice": plan["peak"], "depth": plan["depth"],
          "tflops": 2.0 * plan["flops"] * done / max(seconds, 1e-30) / 1e12, "checks": checks}


def _mera_sliced_run_staged(be, chi, placement, plan, half_dtype, seed, budget_seconds, check_every):
  """mera_sliced_run with the partial contractions that depend on one slice index only computed once per value of
  that index (see there)."""
  import time  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import contractors  # pylint: disable=import-outside-toplevel
  st = _mera_stages(plan)
  nodes = plan["nodes"]
  n_in = len(nodes)
  hk, rk = nodes.index(plan["hnode"]), nodes.index(plan["rnode"])
  steps = {d: [(a, b, new) for a, b, new, dep, _ in st["steps"] if dep == d] for d in ("", "i", "j", "ij")}
  needs = {}
  for d, lst in steps.items():
    produced = {new for _, _, new in lst}
    needs[d] = sorted({x for a, b, _ in lst for x in (a, b)} - produced)
  final_id = st["steps"][-1][2]
  fixed = {}

  def gen(k, i, j):
    shape = st["shapes"][k]
    scale = float(np.prod(shape)) ** -0.25
    sd = seed + 1000 + i if k == hk else (seed + 5000 + j if k == rk else seed + 7 * k)
    return be.device_random(shape, dtype=half_dtype, seed=sd, normal=True, a=0.0, b=scale)

  def input_tensor(k, i, j):
    if k in (hk, rk):
      return gen(k, i, j)
    if k not in fixed:
      fixed[k] = gen(k, 0, 0)          # isometries / disentanglers: slice-independent
    return fixed[k]

  def operands(d, pools, i, j):
    out = {}
    for k in needs[d]:
      if k < n_in:
        out[k] = (input_tensor(k, i, j), st["labels"][k])
      else:
        out[k] = next(pool[k] for pool in pools if k in pool)
    return out

  lt = st["label_time"]
  acc = be.zeros((), dtype=np.float32)
  done, checks, check_seconds = 0, [], 0.0
  be.synchronize()
  t0 = time.perf_counter()
  once = _run_stage(be, operands("", [], 0, 0), steps[""], lt) if steps[""] else {}
  per_i = [_run_stage(be, operands("i", [once], i, 0), steps["i"], lt) if steps["i"] else {} for i in range(chi)]
  be.synchronize()
  seconds_once_per_i = time.perf_counter() - t0
  j_done, stop = 0, False
  for j in range(chi):
    per_j = _run_stage(be, operands("j", [once], 0, j), steps["j"], lt) if steps["j"] else {}
    j_done += 1
    for i in range(chi):
      left = _run_stage(be, operands("ij", [per_i[i], per_j, once], i, j), steps["ij"], lt)
      res = left[final_id][0]
      acc = be.addition(acc, be.reshape(be.cast(res, np.float32), ()))
      done += 1
      if check_every and (done - 1) % check_every == 0:
        be.synchronize()
        tc = time.perf_counter()
        tensors32 = [be.cast(input_tensor(k, i, j), np.float32) for k in range(n_in)]
        real32 = _mera_slice_network(be, plan, lambda k, node, shape: tensors32[k])      # pylint: disable=cell-var-from-loop
        r32 = contractors.contract_path(plan["path"], real32).tensor
        checks.append([[i, j], float(np.asarray(res, dtype=np.float64).reshape(-1)[0]),
                       float(np.asarray(r32, dtype=np.float64).reshape(-1)[0])])
        for nd in real32:
          nd.tensor, nd.edges = None, []
        del real32, r32, tensors32
        be.synchronize()
        check_seconds += time.perf_counter() - tc
      del res, left
      if budget_seconds is not None and done % 16 == 0:
        be.synchronize()
        if time.perf_counter() - t0 - check_seconds > budget_seconds:
          stop = True
          break
    del per_j
    if stop:
      break
  be.synchronize()
  seconds = time.perf_counter() - t0 - check_seconds
  macs = st["macs"]
  executed = macs[""] + chi * macs["i"] + j_done * macs["j"] + done * macs["ij"]
  return {"placement": placement, "n_slices": chi * chi, "slices_done": done, "seconds": seconds,
          "energy_partial_sum": float(np.asarray(acc, dtype=np.float64).reshape(-1)[0]),
          "macs_per_slice": plan["flops"], "peak_elems_per_slice": plan["peak"], "depth": plan["depth"],
          "reuse_partials": True, "executed_macs": executed,
          "seconds_by_phase": {"steps that depend on no index or on i (once per i)": seconds_once_per_i,
                               "steps that depend on j (once per j) and on both (every slice)": seconds - seconds_once_per_i},
          "macs_by_dependence": {"none": macs[""], "i": macs["i"], "j": macs["j"], "ij": macs["ij"]},
          "stage_runs": {"none": 1 if steps[""] else 0, "i": chi, "j": j_done, "ij": done},
          "tflops": 2.0 * executed / max(seconds, 1e-30) / 1e12,
          "tflops_if_every_slice_ran_alone": 2.0 * plan["flops"] * done / max(seconds, 1e-30) / 1e12, "checks": checks}


def ham_ising():
  """3-site critical-Ising term  X Z X - (X X 1 + 1 X X)/2  (Evenbly & White 2016), as used by
  simple_mera.py:298-309."""
  E, X, Z = np.eye(2), np.array([[0., 1.], [1., 0.]]), np.diag([1., -1.])
  k3 = lambda a, b, c: np.kron(a, np.kron(b, c))
  return (k3(X, Z, X) - 0.5 * (k3(X, X, E) + k3(E, X, X))).reshape((2,) * 6)


def wavelet_mera_tensors():
  """D = 2 wavelet MERA (isometry, disentangler) of simple_mera_test.py:91-120, real form.
  The reference writes them with Pauli products; X(x)Y and Y(x)X enter as i*X(x)Y, so with
  iY = [[0, 1], [-1, 0]] everything is real."""
  r2, r3 = np.sqrt(2.0), np.sqrt(3.0)
  E, X, Z = np.eye(2), np.array([[0., 1.], [1., 0.]]), np.diag([1., -1.])
  iY = np.array([[0., 1.], [-1., 0.]])
  wmat = ((r3 + r2) / 4 * np.kron(E, E) + (r3 - r2) / 4 * np.kron(Z, Z)
          + (1 + r2) / 4 * np.kron(X, iY) + (1 - r2) / 4 * np.kron(iY, X))
  umat = ((r3 + 2) / 4 * np.kron(E, E) + (r3 - 2) / 4 * np.kron(Z, Z)
          + 0.25 * np.kron(X, iY) + 0.25 * np.kron(iY, X))
  w = np.transpose(wmat.reshape(2, 2, 2, 2)[:, 0, :, :], (1, 2, 0))
  u = np.transpose(umat.reshape(2, 2, 2, 2), (2, 3, 0, 1))
  return w, u


def mera_random_tensors(chi: int, seed: int = 7, dtype=np.float32):
  """Random isometry / unitary disentangler (QR of Gaussians), Hermitian hamiltonian and a
  PSD unit-trace 3-site state -- the construction of simple_mera_test.py:69-88, real dtype."""
  rng = np.random.default_rng(seed)
  q, _ = np.linalg.qr(rng.standard_normal((chi * chi, chi)))
  iso = q.reshape(chi, chi, chi).astype(dtype)
  q, _ = np.linalg.qr(rng.standard_normal((chi * chi, chi * chi)))
  dis = q.reshape(chi, chi, chi, chi).astype(dtype)
  h = rng.standard_normal((chi**3, chi**3))
  ham = (0.5 * (h + h.T)).reshape((chi,) * 6).astype(dtype)
  r = rng.standard_normal((chi**3, chi**3))
  rho = r @ r.T
  rho = (rho / np.trace(rho)).reshape((chi,) * 6).astype(dtype)
  return ham, rho, iso, dis
