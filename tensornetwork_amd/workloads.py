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
  shape-only tensors, one cut on a leg of the hamiltonian and one on a leg of the state (chi^2 slices), and the path.

  `slice_edge` (reference network_components.py:1636-1682) slices BOTH nodes of an edge: the disentangler / isometry
  at the other end of each cut leg changes with the slice index just as the hamiltonian / state does, so a step of the
  path depends on a cut as soon as it has absorbed EITHER end (`distributed._StagePlan` with windows on both ends).
  What a sliced run executes when every step runs once per distinct value of the cuts it depends on is the cost of
  the same path on the UNSLICED sizes with the two cut legs kept as batch indices -- so that is what the pathfinder
  is asked to minimise: output = {cut legs}, full sizes.  Candidates: all 36 (h leg, state leg) pairs under
  branch(nbranch=2), the four best re-searched with `optimal`; ranked by (executed multiply-adds, peak intermediate
  of a slice, multiply-adds of one slice alone) among the plans whose peak intermediate is at most chi^5 elements.  At chi = 64 every good choice executes 3.72e16 multiply-adds per
  placement -- the flop-optimal DENSE cost of the network: reuse recovers the dense cost, it cannot beat it."""
  import functools  # pylint: disable=import-outside-toplevel
  import itertools  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import distributed, pathfinder  # pylint: disable=import-outside-toplevel
  pb = _PlanBackend()
  ham, rho = _ShapeOnly((chi,) * 6), _ShapeOnly((chi,) * 6)
  plan_nodes = mera_layer_network(pb, ham, rho, _ShapeOnly((chi,) * 3), _ShapeOnly((chi,) * 4), placement)
  inputs = [set(n.edges) for n in plan_nodes]
  sizes = {e: e.dimension for e in network.get_all_edges(plan_nodes)}
  hnode = [n for n in plan_nodes if n.tensor is ham][0]
  rnode = [n for n in plan_nodes if n.tensor is rho][0]
  every = list(itertools.product(range(chi), range(chi)))

  def rate(eh, er, algo):
    path = algo(inputs, {eh, er}, sizes)
    trial = dict(sizes)
    trial[eh] = 1
    trial[er] = 1
    flops, peak = pathfinder.path_cost(inputs, set(), trial, path)
    stage = distributed._StagePlan(plan_nodes, [eh, er], path)      # pylint: disable=protected-access
    return (stage.macs_with_reuse(every), float(peak), float(flops)), path, stage

  branch2 = functools.partial(pathfinder.branch, nbranch=2)
  rows = []
  for a, eh in enumerate(hnode.edges):
    for b, er in enumerate(rnode.edges):
      key, path, stage = rate(eh, er, branch2)
      rows.append((key, (a, b), eh, er, path, stage))
  # no intermediate larger than the sliced rank-6 inputs themselves (chi^5 elements: 2 GB in bf16 at chi = 64; the
  # cheapest-by-a-hair plans keep a chi^6 intermediate -- 137 GB)
  fits = lambda key: key[1] <= float(chi) ** 5
  order = lambda r: (not fits(r[0]), r[0], r[1])
  rows.sort(key=order)
  best = rows[0]
  for _, (a, b), eh, er, _, _ in rows[:4]:
    key, path, stage = rate(eh, er, pathfinder.optimal)
    cand = (key, (a, b), eh, er, path, stage)
    if order(cand) < order(best):
      best = cand
  (executed, peak, flops), _, eh, er, path, stage = best
  return {"nodes": plan_nodes, "hnode": hnode, "rnode": rnode, "cuts": [eh, er], "cut": {id(eh), id(er)}, "path": path,
          "stage": stage, "flops": float(flops), "peak": float(peak), "executed_macs_with_reuse": float(executed),
          "depth": pathfinder.path_depth(path, len(plan_nodes))}


class _Held:
  """what `distributed._contract_slices_staged` needs of an input node: its tensor"""

  def __init__(self, tensor):
    self.tensor = tensor


class MeraSlicedLayer:
  """ONE placement of the binary-MERA layer energy (examples/simple_mera/simple_mera.py:54-112) as the network
  `slice_edge(cut_h, i, 1); slice_edge(cut_rho, j, 1)` leaves of it, for every (i, j), at a bond dimension whose rank-6
  tensors cannot be materialised (chi = 64: 137 GB each).

  The layer's tensors are DEFINED, not stored: the hamiltonian slice-wise along its cut leg, h[.., i, ..] = R(seed_h + i)
  (R a counter-based generator of the sliced shape), the state likewise along its own, rho[.., j, ..] = R(seed_r + j);
  ONE isometry (chi^3) and ONE disentangler (chi^4) as full tensors shared by all their nodes, as in the reference.
  Slice (i, j) of the network is then: h's slice i, rho's slice j, the disentangler / isometry at the OTHER end of
  each cut leg sliced at the same index (slice_edge slices both nodes of the edge), every other node whole.  The
  slices are contracted by `distributed._contract_slices_staged` -- the machinery `contract_sliced` runs -- with the
  two rank-6 inputs handed in slice by slice (`input_provider`).  `host_tensors()` materialises the same network on
  the host for small chi (tests: the slices must add up to the dense energy)."""

  def __init__(self, be, chi: int, placement: str, dtype, seed: int = 40):
    self.be, self.chi, self.placement, self.dtype, self.seed = be, int(chi), placement, dtype, seed
    self.plan = _mera_slice_plan(self.chi, placement)
    self.stage = self.plan["stage"]
    nodes = self.plan["nodes"]
    self.hk, self.rk = nodes.index(self.plan["hnode"]), nodes.index(self.plan["rnode"])
    self.rank = [len(n.edges) for n in nodes]
    self.scale = {3: float(chi) ** -0.75, 4: float(chi) ** -1.0, 6: float(chi) ** -1.5}      # closed network: variance 1
    self._full = {}          # rank -> full tensor (isometry, disentangler) by dtype name
    # axis of the cut leg on the two virtual inputs
    self.h_axis = [ax for ax, c in self.stage.windows[self.hk] if c == 0][0]
    self.r_axis = [ax for ax, c in self.stage.windows[self.rk] if c == 1][0]

  # ---- tensors
  def _virtual(self, k, index, dtype):
    axis = self.h_axis if k == self.hk else self.r_axis
    shape = tuple(1 if ax == axis else self.chi for ax in range(6))
    sd = self.seed + (1000 if k == self.hk else 5000) + int(index)
    return self.be.device_random(shape, dtype=self.dtype, seed=sd, normal=True, a=0.0, b=self.scale[6]) \
        if dtype is None else self.be.cast(
            self.be.device_random(shape, dtype=self.dtype, seed=sd, normal=True, a=0.0, b=self.scale[6]), dtype)

  def _shared(self, rank, dtype):
    key = (rank, str(dtype))
    if key not in self._full:
      t = self.be.device_random((self.chi,) * rank, dtype=self.dtype, seed=self.seed + 7 * rank, normal=True, a=0.0,
                                b=self.scale[rank])
      self._full[key] = t if dtype is None else self.be.cast(t, dtype)
    return self._full[key]

  def holders(self, dtype=None):
    """inputs as `_contract_slices_staged` wants them: full tensors for the isometry / disentangler nodes, None for the
    two rank-6 nodes (handed in slice by slice).  dtype: None = the layer's own, else a cast of the same values"""
    return [_Held(None if k in (self.hk, self.rk) else self._shared(self.rank[k], dtype)) for k in range(len(self.rank))]

  def provider(self, dtype=None):
    def give(k, idx):
      if k == self.hk:
        return self._virtual(k, idx[0], dtype)
      if k == self.rk:
        return self._virtual(k, idx[1], dtype)
      return None          # a real input with a cut leg: sliced from its full tensor by the caller
    return give

  def host_tensors(self):
    """(hamiltonian, state, isometry, disentangler) of the SAME layer as host float64 arrays (small chi only)"""
    if self.chi > 12:
      raise ValueError("the rank-6 tensors are materialised for tests at small chi only")
    cat = lambda k, axis: np.concatenate([np.asarray(self._virtual(k, i, None), dtype=np.float64) for i in range(self.chi)],
                                         axis=axis)
    return (cat(self.hk, self.h_axis), cat(self.rk, self.r_axis),
            np.asarray(self._shared(3, None), dtype=np.float64), np.asarray(self._shared(4, None), dtype=np.float64))

  def all_slices(self):
    import itertools  # pylint: disable=import-outside-toplevel
    return self.stage.ordered(list(itertools.product(range(self.chi), range(self.chi))))

  # ---- contraction
  def contract(self, slices, reuse=True, dtype=None, stats=None, on_slice=None, partials_out=None):
    """sum of the partial energies of `slices` (a device scalar, f32-accumulated); reuse=False: every step in every
    slice (the slice-by-slice run)"""
    from tensornetwork_amd import distributed  # pylint: disable=import-outside-toplevel
    total, _ = distributed._contract_slices_staged(      # pylint: disable=protected-access
        self.be, self.holders(dtype), self.stage, slices, None, partials_out, stats, input_provider=self.provider(dtype),
        on_slice=on_slice, reuse=reuse)
    return total


def mera_sliced_sample(be, chi: int, dtype, reps: int = 2, seed: int = 17):
  """Seconds of ONE slice contracted on its own, per placement (the EXTRAPOLATED rows of the bench: x slice count):
  slice (0, 0) of `MeraSlicedLayer` -- every step of the path, nothing reused -- best of `reps` after a warm-up."""
  import time  # pylint: disable=import-outside-toplevel
  out = {}
  for placement in ("left", "right"):
    layer = MeraSlicedLayer(be, chi, placement, dtype, seed=seed)
    best_t = None
    for _ in range(reps + 1):
      be.synchronize()
      t0 = time.perf_counter()
      res = layer.contract([(0, 0)], reuse=False)
      be.synchronize()
      t = time.perf_counter() - t0
      best_t = t if best_t is None else min(best_t, t)
      del res
    plan = layer.plan
    out[placement] = {"n_slices": chi * chi, "macs_per_slice": plan["flops"], "peak_elems_per_slice": plan["peak"],
                      "sec_per_slice": best_t, "tflops": 2.0 * plan["flops"] / best_t / 1e12}
  return out


def mera_slice_values(be, chi: int, slices, half_dtype, seed: int = 40):
  """Slices (i, j) of the layer `MeraSlicedLayer` defines, each contracted on its own in `half_dtype` and in f32 on
  the same (half-rounded) values.  Returns {placement: {"depth": d, "rows": [{"slice": (i, j), "half": .., "f32": ..}]}}."""
  out = {}
  for placement in ("left", "right"):
    layer = MeraSlicedLayer(be, chi, placement, half_dtype, seed=seed)
    rows = []
    for (i, j) in slices:
      vals = {}
      for name, dt in (("half", None), ("f32", np.float32)):
        res = layer.contract([(int(i), int(j))], reuse=False, dtype=dt)
        vals[name] = float(np.asarray(res, dtype=np.float64).reshape(-1)[0])
        del res
      rows.append({"slice": [int(i), int(j)], **vals})
    out[placement] = {"depth": layer.plan["depth"], "rows": rows}
  return out


def mera_sliced_run(be, chi: int, placement: str, half_dtype, seed: int = 40, budget_seconds: float = None,
                    check_every: int = 0, reuse_partials: bool = True):
  """ONE placement of the bond-sliced binary-MERA layer energy as a RUN: all chi^2 slices of `MeraSlicedLayer` (what
  `slice_edge` on one leg of the hamiltonian and one leg of the state leaves of the layer, BOTH ends of each cut leg
  sliced) through `distributed._contract_slices_staged`, partial energies added in f32 on the device (one read-back).

  reuse_partials (default): every step runs once per distinct value of the cuts it depends on; the results of a class
  of steps whose values come back are kept for all of them when they fit the cache budget (60 % of the free HBM; chi =
  64: the class that depends on the FAST cut alone -- 64 results of 2 GB -- stays, the slow cut's class changes once per
  64 slices and keeps the result in use only; the class that depends on both is 98 % of the executed multiply-adds and
  runs in every slice).  False: every step in every slice.  `executed_macs` counts what RAN.

  budget_seconds: stop after the first slice that ends beyond it (checked every 16 slices; the record says how many
  slices ran).  check_every: every that many slices the same slice is also contracted alone in f32 on the same
  half-rounded values (`checks`: [(i, j), half, f32]) -- a sample of the a-priori rounding check, outside the timing."""
  import time  # pylint: disable=import-outside-toplevel
  layer = MeraSlicedLayer(be, chi, placement, half_dtype, seed=seed)
  plan, stage = layer.plan, layer.stage
  slices = layer.all_slices()
  checks, clock = [], {"check": 0.0}
  be.synchronize()
  t0 = time.perf_counter()

  def on_slice(done, idx, tensor):
    if check_every and (done - 1) % check_every == 0:
      be.synchronize()
      tc = time.perf_counter()
      r32 = layer.contract([idx], reuse=False, dtype=np.float32)
      checks.append([[int(idx[0]), int(idx[1])], float(np.asarray(tensor, dtype=np.float64).reshape(-1)[0]),
                     float(np.asarray(r32, dtype=np.float64).reshape(-1)[0])])
      be.synchronize()
      clock["check"] += time.perf_counter() - tc
    if budget_seconds is not None and done % 16 == 0:
      be.synchronize()
      return time.perf_counter() - t0 - clock["check"] > budget_seconds
    return False

  stats = {}
  acc = layer.contract(slices, reuse=reuse_partials, stats=stats, on_slice=on_slice)
  be.synchronize()
  seconds = time.perf_counter() - t0 - clock["check"]
  done = int(stats["slices_done"])
  name = {"-": "none", "0": "i", "1": "j", "0,1": "ij"}
  macs = {name[",".join(str(k) for k in sorted(c)) or "-"]: m for c, m in stage.class_macs.items()}
  runs = {name[k]: v for k, v in stats["stage_runs"].items()}
  for k in ("none", "i", "j", "ij"):
    macs.setdefault(k, 0.0)
    runs.setdefault(k, 0)
  executed = float(stats["executed_macs"])
  return {"placement": placement, "n_slices": chi * chi, "slices_done": done, "seconds": seconds,
          "energy_partial_sum": float(np.asarray(acc, dtype=np.float64).reshape(-1)[0]),
          "macs_per_slice": plan["flops"], "peak_elems_per_slice": plan["peak"], "depth": plan["depth"],
          "reuse_partials": bool(reuse_partials), "executed_macs": executed,
          "model_macs_with_reuse_all_slices": plan["executed_macs_with_reuse"],
          "macs_by_dependence": macs, "stage_runs": runs,
          "classes_kept_for_all_values": stats.get("classes_kept_for_all_values"),
          "semantics": "slice_edge on both ends of each cut leg (the disentangler / isometry next to a cut leg is sliced "
                       "with it); hamiltonian and state defined slice-wise, never materialised",
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
