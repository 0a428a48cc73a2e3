"""Multi-GPU contraction by bond slicing (one process per GPU, RCCL over xGMI).

The reference has no distributed path at all; its own primitive for restricting
a bond to a sub-range is ``slice_edge`` (``network_components.py:1636-1682``).
Fixing the values of ``s`` contracted bonds of dimensions ``d_1..d_s`` turns one
network into ``d_1*...*d_s`` independent networks of identical topology whose
results add up.  That is the natural partition of a contractor path:

  * the pairwise order is searched ONCE on the sliced topology (host),
  * slices are dealt to the ranks (contiguous blocks of the order in which the costly cut bonds vary slowest;
    round-robin in the slice-by-slice mode); every rank contracts its slices with the same path on its own GPU --
    each step once per distinct value of the cut bonds it depends on -- and accumulates the partial result locally,
  * ONE all-reduce(sum) of the (small) result tensor finishes the job -- the
    only data-path collective, and only because the partition has a genuine
    exchange step.  On GPUs it is RCCL through libtnhip's own K8 entry points
    (``tensornetwork_amd.comm.RcclComm``: ``tnh_allreduce`` on the library stream) -- the only
    communicator of the package.  The CPU test-suite drives the same partitioning code with the
    oracle backend over gloo through a TEST-side adapter (``tests/gloo_comm.py``: any object with
    ``rank`` / ``world`` / ``all_reduce_sum`` / ``all_gather_rows`` / ``all_gather_counts`` will do).

Networks that do not partition (MPS chains zipped from a boundary, a single
SVD) are run as independent replicas instead -- see bench.py.
"""
import itertools
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from tensornetwork_amd import contractors, network, pathfinder


# ------------------------------------------------------------------ collectives
class LocalComm:
  """Single-process stand-in (world size 1)."""
  rank, world = 0, 1

  def all_reduce_sum(self, backend, tensor):  # pylint: disable=unused-argument
    return tensor

  def all_gather_rows(self, backend, tensor, rows_per_rank):  # pylint: disable=unused-argument
    return tensor


# ---------------------------------------------------------------- slice planning
def _index_problem(nodes: Sequence[network.Node]):
  inputs = [set(n.edges) for n in nodes]
  output = network.get_subgraph_dangling(nodes)
  sizes = {e: e.dimension for e in network.get_all_edges(nodes)}
  return inputs, output, sizes


def choose_cut_edges(nodes: Sequence[network.Node], min_slices: int,
                     algorithm: Callable = pathfinder.greedy, beam: Optional[int] = None,
                     world: int = 1) -> List[network.Edge]:
  """Pick contracted edges to slice until there are >= min_slices slices.

  The cut SET is ranked by what `contract_sliced` then executes (`_StagePlan`): first the estimated time
  (`step_seconds`: a flop / byte / launch roofline per pairwise step) of the SLOWEST of `world` ranks under the
  partition `contract_sliced` uses (`_StagePlan.partition`) -- every step once per distinct value of the cuts it
  depends on -- then the sum over the ranks, then the multiply-adds, then the peak intermediate.  Two searches, the better set wins (ties: the sequential one):

    * the sequential rule: cut the edge whose removal (dimension -> 1) gives the cheapest re-searched path, repeat;
    * a beam search over cut sets (`beam` partial sets per level, each extended by one of the `beam` best single
      cuts).  `beam=None` (default): 24 for networks of at most 160 candidate edges (the 64-node 3-regular network has
      96: 2.6 s of host time), else off; `beam=0`: the sequential rule alone (rounds 1-4).

  Every rank must call this with the same arguments: the search is deterministic (candidates in the order of their
  first end's position in `nodes`, strict comparisons)."""
  nodes = list(nodes)
  seq = _choose_cut_edges_sequential(nodes, min_slices, algorithm)
  _, _, sizes = _index_problem(nodes)
  n_cand = sum(1 for e in sizes if not e.is_dangling() and not e.is_trace() and sizes[e] > 1)
  if beam is None:
    beam = 24 if n_cand <= 160 else 0
  if beam <= 0 or not seq or any(e.is_trace() for n in nodes for e in n.edges if not e.is_dangling()):
    return seq
  cost = _cut_cost_fn(nodes, algorithm, max(int(world), 1))
  found = _choose_cut_edges_beam(nodes, min_slices, beam, cost)
  if found is None:
    return seq
  return found if cost(found) < cost(seq) else seq


def _cut_cost_fn(nodes, algorithm, world):
  """cost(cuts) -> (slowest rank's estimated seconds with reuse, all ranks', multiply-adds with reuse, peak intermediate)"""
  inputs, output, sizes = _index_problem(nodes)
  memo = {}

  def cost(cuts):
    ident = frozenset(id(e) for e in cuts)
    if ident not in memo:
      trial = dict(sizes)
      for e in cuts:
        trial[e] = 1
      path = algorithm(inputs, output, trial)
      plan = _StagePlan(nodes, list(cuts), path)
      flops, peak = pathfinder.path_cost(inputs, output, trial, path)
      every = list(itertools.product(*[range(sizes[e]) for e in cuts]))
      per_rank = [plan.seconds_with_reuse(b) for b in plan.partition(every, world) if b]
      memo[ident] = (max(per_rank), sum(per_rank), plan.macs_with_reuse(every), float(peak))
    return memo[ident]
  return cost


def _choose_cut_edges_sequential(nodes, min_slices, algorithm):
  inputs, output, sizes = _index_problem(nodes)
  sizes = dict(sizes)
  cuts: List[network.Edge] = []
  n_slices = 1
  # deterministic candidate order (every rank must pick the same cuts): by the
  # position of the edge's first end in `nodes`
  pos = {id(n): i for i, n in enumerate(nodes)}
  def edge_key(e):
    return min((pos[id(nd)], ax) for nd, ax in e.ends())
  candidates = sorted((e for e in sizes if e not in output and not e.is_trace() and sizes[e] > 1),
                      key=edge_key)
  while n_slices < min_slices and candidates:
    best = None
    for e in candidates:
      trial = dict(sizes)
      trial[e] = 1
      path = algorithm(inputs, output, trial)
      flops, peak = pathfinder.path_cost(inputs, output, trial, path)
      key = (flops * sizes[e], peak)   # total work over all slices of this edge, then memory
      if best is None or key < best[0]:
        best = (key, e)   # strict '<': first candidate wins ties
    e = best[1]
    cuts.append(e)
    n_slices *= sizes[e]
    sizes[e] = 1
    candidates.remove(e)
  return cuts


def _choose_cut_edges_beam(nodes, min_slices, beam, cost):
  inputs, output, sizes = _index_problem(nodes)
  pos = {id(n): i for i, n in enumerate(nodes)}
  key_of = lambda e: min((pos[id(nd)], ax) for nd, ax in e.ends())
  candidates = sorted((e for e in sizes if e not in output and not e.is_trace() and sizes[e] > 1), key=key_of)
  singles = sorted(((cost([e]), key_of(e), e) for e in candidates), key=lambda t: (t[0], t[1]))
  pool = [t[2] for t in singles[:beam]]
  partial = [(c, [e]) for c, _, e in singles[:beam]]
  while partial:
    done = [p for p in partial if int(np.prod([sizes[e] for e in p[1]])) >= min_slices]
    if done:
      return min(done, key=lambda p: (p[0], [key_of(e) for e in p[1]]))[1]
    grown, seen = [], set()
    for _, cuts in partial:
      for e in pool:
        if e in cuts:
          continue
        ident = frozenset(id(x) for x in cuts + [e])
        if ident in seen:
          continue
        seen.add(ident)
        new = sorted(cuts + [e], key=key_of)
        grown.append((cost(new), new))
    if not grown:
      break
    grown.sort(key=lambda p: (p[0], [key_of(e) for e in p[1]]))
    partial = grown[:beam]
  return None     # (fewer candidates than levels: the sequential rule stands)


def _variant_steps(n_inputs: int, variant_inputs: Sequence[bool], path):
  """Walk a linear path symbolically: [(id_a, id_b, id_new, variant)] per pairwise step, ids 0 .. n_inputs - 1 for the
  inputs and n_inputs + s for the result of step s; a result is variant when one of its operands is."""
  ids = list(range(n_inputs))
  variant = list(variant_inputs)
  steps = []
  for pair in path:
    if len(pair) == 1:
      continue
    a, b = sorted(pair)
    ia, ib, new = ids[a], ids[b], n_inputs + len(steps)
    variant.append(variant[ia] or variant[ib])
    steps.append((ia, ib, new, variant[new]))
    ids = [x for k, x in enumerate(ids) if k not in (a, b)] + [new]
  return steps


def _hoist_invariant(nodes, cut_edges, path, output_edge_order):
  """Every slice contracts the same network with the same path; a pairwise step neither of whose operands
  depends on a cut bond gives the same tensor in every slice.  Those steps are run ONCE here, on a copy of
  the network, and the slices contract what is left: the inputs that touch a cut bond, the slice-invariant
  intermediates, and the remaining steps of the SAME path in the same order (so every slice partial is the
  tensor it was before).  Returns (nodes, cut_edges, path, output_edge_order, steps hoisted) of the reduced
  network, or None when there is nothing to hoist."""
  if any(e.is_trace() for n in nodes for e in n.edges if not e.is_dangling()):
    return None       # (contract_path folds trace edges first, which renumbers nothing but replaces nodes)
  touched = {id(nd) for e in cut_edges for nd, _ in e.ends()}
  steps = _variant_steps(len(nodes), [id(n) in touched for n in nodes], path)
  hoist = [st for st in steps if not st[3]]
  if not hoist:
    return None
  node_map, edge_map = network.copy(nodes)
  by_id = {k: node_map[n] for k, n in enumerate(nodes)}
  # layout planning as contract_path does it, with the times of the WHOLE path: an invariant intermediate is laid
  # out for the step that consumes it, once, instead of being permuted in every slice
  edge_time = contractors._edge_times(path, [node_map[n] for n in nodes])      # pylint: disable=protected-access
  for ia, ib, new, _ in hoist:
    by_id[new] = network.contract_between(by_id.pop(ia), by_id.pop(ib), allow_outer_product=True, edge_time=edge_time)
  live = sorted(by_id)                      # what the slices start from
  cur, rest = list(live), []
  for ia, ib, new, is_variant in steps:
    if not is_variant:
      continue
    rest.append(tuple(sorted((cur.index(ia), cur.index(ib)))))
    cur = [x for x in cur if x not in (ia, ib)] + [new]
  order = [edge_map[e] for e in output_edge_order] if output_edge_order is not None else None
  return [by_id[k] for k in live], [edge_map[e] for e in cut_edges], rest, order, len(hoist)


# Roofline estimate of one pairwise step on an MI355X, used to RANK cut sets and partitions (never reported as a
# measurement): the larger of the multiply-adds at 600 TFLOP/s and the operand + result bytes (2-byte elements) at
# 2.5 TB/s, plus 8 us of launches.  Calibrated on the D = 12 64-node network (profiles/r05_rr_scaling_rehearsal.jsonl):
# the cuts with the FEWEST multiply-adds (1.73e13 against 2.18e13) ran 3.4x SLOWER on the GPU (0.312 s against 0.093 s)
# because 69 % of their work sits in thin per-slice products that move bytes, not flops; this model ranks the two sets
# the right way round (0.18 s against 0.11 s), a multiply-add count does not.
STEP_FLOPS_PER_SECOND = 6.0e14
STEP_BYTES_PER_SECOND = 2.5e12
STEP_LAUNCH_SECONDS = 8.0e-6


def step_seconds(macs: float, elements_moved: float) -> float:
  return max(2.0 * macs / STEP_FLOPS_PER_SECOND, 2.0 * elements_moved / STEP_BYTES_PER_SECOND) + STEP_LAUNCH_SECONDS


class _StagePlan:
  """Which steps of a sliced network's path depend on which cut bonds.

  An input depends on the cuts it touches, a step on the union of its operands' cuts.  A step that depends on the
  cuts S gives the same tensor in all slices that agree on the index values of S: it needs to run once per value
  tuple of S, not once per slice.  (The D = 12 north-star network with its two cuts: 38 of 63 steps depend on no cut,
  20 steps -- 99.4 % of a slice's multiply-adds -- on the second cut alone, 5 small ones on the first or on both:
  12 runs of the expensive part instead of 144.)"""

  def __init__(self, nodes, cut_edges, path):
    self.n = len(nodes)
    self.dims = [e.dimension for e in cut_edges]
    inputs, output, sizes = _index_problem(nodes)
    sliced = dict(sizes)
    for e in cut_edges:
      sliced[e] = 1
    pos = {id(nd): k for k, nd in enumerate(nodes)}
    self.windows: Dict[int, List[Tuple[int, int]]] = {}          # input -> [(axis, cut number)]
    for c, e in enumerate(cut_edges):
      for nd, ax in e.ends():
        self.windows.setdefault(pos[id(nd)], []).append((ax, c))
    dep = [frozenset(c for _, c in self.windows.get(k, [])) for k in range(self.n)]
    ids = list(range(self.n))
    remaining = [frozenset(x) for x in inputs]
    self.steps = []                                              # (id_a, id_b, id_new, cuts it depends on, multiply-adds)
    self.step_seconds = []                                       # roofline estimate per step (step_seconds)
    for pair in path:
      if len(pair) == 1:
        continue
      a, b = sorted(pair)
      k1, k2 = remaining[a], remaining[b]
      others = set(output)
      for t, k in enumerate(remaining):
        if t not in (a, b):
          others |= k
      new = self.n + len(self.steps)
      dep.append(dep[ids[a]] | dep[ids[b]])
      kept = frozenset(d for d in (k1 | k2) if d in others)
      size = lambda ks: float(pathfinder._size(ks, sliced))      # pylint: disable=protected-access
      self.steps.append((ids[a], ids[b], new, dep[new], size(k1 | k2)))
      self.step_seconds.append(step_seconds(size(k1 | k2), size(k1) + size(k2) + size(kept)))
      ids = [x for t, x in enumerate(ids) if t not in (a, b)] + [new]
      remaining = [k for t, k in enumerate(remaining) if t not in (a, b)] + [kept]
    self.dep = dep
    self.final = self.steps[-1][2] if self.steps else 0
    self.classes = sorted({st[3] for st in self.steps}, key=lambda c: (len(c), sorted(c)))
    self.class_steps = {c: [(a, b, new) for a, b, new, d, _ in self.steps if d == c] for c in self.classes}
    self.class_macs = {c: sum(st[4] for st in self.steps if st[3] == c) for c in self.classes}
    self.class_seconds = {c: sum(t for st, t in zip(self.steps, self.step_seconds) if st[3] == c) for c in self.classes}
    self.class_needs = {}
    for c, lst in self.class_steps.items():
      produced = {new for _, _, new in lst}
      self.class_needs[c] = sorted({x for a, b, _ in lst for x in (a, b)} - produced)
    self.labels = [[id(e) for e in nd.edges] for nd in nodes]
    self.label_time = {id(e): t for e, t in contractors._edge_times(path, nodes).items()}     # pylint: disable=protected-access
    # loop order of the slices: the cut whose steps cost most varies slowest
    weight = [sum(m for c, m in self.class_seconds.items() if k in c) for k in range(len(cut_edges))]
    self.loop_order = sorted(range(len(cut_edges)), key=lambda k: (-weight[k], k))

  def macs_alone(self) -> float:
    return sum(st[4] for st in self.steps)

  def macs_with_reuse(self, slices) -> float:
    """multiply-adds executed for `slices` (index tuples) when every class runs once per distinct value tuple"""
    total = 0.0
    for c, macs in self.class_macs.items():
      total += macs * len({tuple(idx[k] for k in sorted(c)) for idx in slices})
    return total

  def seconds_with_reuse(self, slices) -> float:
    """`step_seconds` summed over what runs for `slices`: every class once per distinct value tuple"""
    total = 0.0
    for c, sec in self.class_seconds.items():
      total += sec * len({tuple(idx[k] for k in sorted(c)) for idx in slices})
    return total

  def ordered(self, slices, loop_order=None):
    order = self.loop_order if loop_order is None else loop_order
    return sorted(slices, key=lambda idx: tuple(idx[k] for k in order))

  def partition(self, slices, world: int):
    """The slices dealt to `world` ranks.  Candidates: (1) contiguous blocks of the slices sorted with one cut varying
    slowest, either balanced (sizes differ by at most one) or of ceil(n / world) slices each, for all loop orders (at
    most 24 tried, the weight order first); (2) when `slices` is the full product of the cuts' ranges, GRIDS: world =
    w_0 x w_1 x ..., cut k's range split into w_k balanced sub-ranges, one rank per cell (a rank then repeats only
    d_k / w_k values of every single-cut class and keeps as many results: the chi = 64 MERA placement on 8 ranks as
    4 x 2 keeps 32 results of 2 GB per rank instead of 64).  Chosen: the candidate whose SLOWEST rank has the smallest
    estimated time (`seconds_with_reuse`), then the smallest sum, then the earlier candidate (a grid only replaces
    contiguous blocks when it is strictly better).  Pure host arithmetic on the plan: every rank computes the same
    blocks.  A rank's block may be empty (fewer slices than ranks)."""
    slices = list(slices)
    world = max(int(world), 1)
    if world == 1:
      return [self.ordered(slices)]
    orders = [tuple(self.loop_order)]
    for perm in itertools.permutations(range(len(self.dims))):
      if perm not in orders and len(orders) < 24:
        orders.append(perm)
    best = None
    for order in orders:
      seq = self.ordered(slices, order)
      base, extra = divmod(len(seq), world)
      per = -(-len(seq) // world)
      bounds_balanced, start = [], 0
      for r in range(world):
        stop = start + base + (1 if r < extra else 0)
        bounds_balanced.append((start, stop))
        start = stop
      for bounds in (bounds_balanced, [(r * per, (r + 1) * per) for r in range(world)]):
        blocks = [seq[a:b] for a, b in bounds]
        loads = [self.seconds_with_reuse(b) for b in blocks]
        key = (max(loads), sum(loads))
        if best is None or key < best[0]:
          best = (key, blocks)
    full = len(self.dims) > 1 and len(slices) == int(np.prod(self.dims)) and \
        set(slices) == set(itertools.product(*[range(d) for d in self.dims]))
    if full:
      for grid in self._grids(world):
        if sum(1 for w in grid if w > 1) < 2:
          continue                     # one cut split only: the contiguous blocks above
        ranges = []
        for d, w in zip(self.dims, grid):
          base, extra = divmod(d, w)
          edges = [0]
          for r in range(w):
            edges.append(edges[-1] + base + (1 if r < extra else 0))
          ranges.append([range(edges[r], edges[r + 1]) for r in range(w)])
        blocks = [self.ordered(list(itertools.product(*cell))) for cell in itertools.product(*ranges)]
        loads = [self.seconds_with_reuse(b) for b in blocks]
        key = (max(loads), sum(loads))
        if key < best[0]:
          best = (key, blocks)
    return best[1]

  def _grids(self, world: int):
    """factorisations of `world` over the cuts, factor k at most the cut's dimension"""
    out = []

    def rec(k, left, acc):
      if k == len(self.dims):
        if left == 1:
          out.append(tuple(acc))
        return
      for w in range(1, min(left, self.dims[k]) + 1):
        if left % w == 0:
          rec(k + 1, left // w, acc + [w])
    rec(0, world, [])
    return out


def _itemsize(t) -> int:
  dt = str(getattr(t, "dtype", "float32"))
  if dt in ("bfloat16", "float16"):
    return 2
  try:
    return int(np.dtype(dt).itemsize)
  except TypeError:
    return 4


# Results of a class of steps whose value tuple comes BACK after others have intervened (a class keyed on a cut that
# is not the slowest of the loop nest) are kept for all its values when they fit the cache budget: 60 % of the HBM that
# is free when the rank's slice loop starts on a hip backend (the chi = 64 MERA placement keeps 64 results of 2 GB --
# 137 GB -- for the fast cut; recomputing them instead cost 48 % more executed work: profiles/r05_bench_trip3.json),
# STAGE_CACHE_BYTES on any other backend.  A class whose value tuples are visited in one contiguous run each keeps
# the result in use only.
STAGE_CACHE_BYTES = 16 << 30


def _cache_budget(be) -> int:
  lib = getattr(be, "lib", None)
  try:
    import ctypes  # pylint: disable=import-outside-toplevel
    name = ctypes.create_string_buffer(64)
    cus, hbm = ctypes.c_int(0), ctypes.c_int64(0)
    in_use, cached, peak = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    if lib.tnh_device_info(name, 64, ctypes.byref(cus), ctypes.byref(hbm)) == 0 and \
        lib.tnh_mem_stats(ctypes.byref(in_use), ctypes.byref(cached), ctypes.byref(peak)) == 0 and hbm.value > 0:
      return max(int(0.6 * (hbm.value - in_use.value)), STAGE_CACHE_BYTES // 16)
  except Exception:  # pylint: disable=broad-except
    pass
  return STAGE_CACHE_BYTES


def _contract_slices_staged(be, nodes, plan: _StagePlan, slices, output_edge_order, partials_out, stats,
                            input_provider=None, cache_bytes=None, on_slice=None, reuse=True):
  """The slices of one rank with every step run once per distinct value of the cuts it depends on (`_StagePlan`).
  Results of a class are kept for the value tuple in use, and for all value tuples when they fit in `cache_bytes`
  (default STAGE_CACHE_BYTES, counted with the tensors' real item size); otherwise the class is recomputed when its
  value tuple comes back (`stats["executed_macs"]` counts what RAN, not the model).  Every slice partial is the tensor
  the slice-by-slice contraction gives: same steps, same order, same operands.

  `input_provider(k, idx)` (optional): tensor of input k for slice idx WITH its cut legs at extent 1, or None to
  slice `nodes[k].tensor` here -- for inputs that are defined slice-wise and never materialised (the 137 GB rank-6
  tensors of the chi = 64 MERA layer).  `on_slice(n_done, idx, tensor)`: called after every slice with its partial
  (time budgets, sampled checks); returning True stops the loop.  `reuse=False`: nothing is kept from one slice to
  the next -- every step runs in every slice (the slice-by-slice baseline on the same machinery)."""
  budget = {"left": _cache_budget(be) if cache_bytes is None else int(cache_bytes)}
  slices = list(slices)
  # which classes see a value tuple again after another one intervened (only those need more than the result in use)
  revisited, distinct = {}, {}
  for c in plan.classes:
    keys = [tuple(idx[k] for k in sorted(c)) for idx in slices]
    runs_of_keys = sum(1 for t in range(len(keys)) if t == 0 or keys[t] != keys[t - 1])
    distinct[c] = len(set(keys))
    revisited[c] = runs_of_keys > distinct[c]
  n = plan.n
  cache: Dict[frozenset, Dict[tuple, dict]] = {c: {} for c in plan.classes}
  keep_all: Dict[frozenset, bool] = {}
  runs = {c: 0 for c in plan.classes}
  sliced_inputs: Dict[int, Tuple[tuple, object]] = {}

  def input_tensor(k, idx):
    wins = plan.windows.get(k)
    if not wins:
      return nodes[k].tensor
    key = tuple(idx[c] for _, c in wins)
    if k in sliced_inputs and sliced_inputs[k][0] == key:
      return sliced_inputs[k][1]
    if input_provider is not None:
      given = input_provider(k, idx)
      if given is not None:
        sliced_inputs[k] = (key, given)
        return given
    t = nodes[k].tensor
    shape = list(be.shape_tuple(t))
    starts = [0] * len(shape)
    for ax, c in wins:
      starts[ax], shape[ax] = idx[c], 1
    sliced_inputs[k] = (key, be.slice(t, tuple(starts), tuple(shape)))
    return sliced_inputs[k][1]

  def stage(c, idx):
    key = tuple(idx[k] for k in sorted(c))
    got = cache[c].get(key)
    if got is not None:
      return got
    operands = {}
    for x in plan.class_needs[c]:
      operands[x] = (input_tensor(x, idx), plan.labels[x]) if x < n else stage(plan.dep[x], idx)[x]
    out = contractors.contract_labelled(be, operands, plan.class_steps[c], plan.label_time)
    runs[c] += 1
    if c not in keep_all:
      nbytes = sum(int(np.prod(be.shape_tuple(t))) * _itemsize(t) for t, _ in out.values())
      need = distinct[c] * nbytes
      keep_all[c] = bool(reuse and revisited[c] and need <= budget["left"])
      if keep_all[c]:
        budget["left"] -= need
    if not keep_all[c]:
      cache[c].clear()
    cache[c][key] = out
    return out

  want = [id(e) for e in output_edge_order] if output_edge_order is not None else None
  total, narrow, done = None, None, 0
  try:
    for idx in slices:
      tensor, labels = stage(plan.dep[plan.final], idx)[plan.final]
      if want is None:
        want = list(labels)      # no order asked for: the first slice's (the label order a backend returns may depend
                                 # on operand alignment; every later partial is brought to the same one before it is added)
      if list(labels) != want:
        tensor = be.transpose(tensor, tuple(list(labels).index(lab) for lab in want))
      part, narrow = _widen(be, tensor)
      if partials_out is not None:
        partials_out.append(np.asarray(part, dtype=np.float64).copy())
      total = be.multiply(part, 1.0) if total is None else be.addition(total, part)
      done += 1
      if not reuse:
        for kept in cache.values():
          kept.clear()
        sliced_inputs.clear()
      if on_slice is not None and on_slice(done, idx, tensor):
        break
    if stats is not None:
      stats["stage_runs"] = {",".join(str(k) for k in sorted(c)) or "-": r for c, r in runs.items()}
      stats["executed_macs"] = sum(plan.class_macs[c] * r for c, r in runs.items())
      stats["slices_done"] = done
      stats["classes_kept_for_all_values"] = {",".join(str(k) for k in sorted(c)) or "-": bool(v) for c, v in keep_all.items()}
      stats["classes_revisited"] = {",".join(str(k) for k in sorted(c)) or "-": bool(v) for c, v in revisited.items()}
    return total, narrow
  finally:
    # `stage` refers to itself (a reference cycle through its closure): without this the kept results -- up to 60 % of
    # the HBM -- would stay allocated until the cyclic collector runs, and the next call would size its cache on what is left
    for kept in cache.values():
      kept.clear()
    sliced_inputs.clear()


def contract_sliced(nodes: Sequence[network.Node], cut_edges: Sequence[network.Edge],
                    comm=None, algorithm: Callable = pathfinder.greedy,
                    output_edge_order: Optional[Sequence[network.Edge]] = None,
                    use_graph: Optional[bool] = None, partials_out: Optional[list] = None,
                    hoist_invariant: bool = True, stats: Optional[dict] = None, reuse: Optional[bool] = None):
  """Contract `nodes` by summing over all index values of `cut_edges`.

  Returns the backend tensor of the full contraction (identical on every rank).
  `output_edge_order` refers to edges of the ORIGINAL network.

  `use_graph` (default: automatically, when the backend offers ``capture`` and this rank
  has >= 4 slices): every slice runs the same launch sequence on tensors of the same
  shape, so the path is captured ONCE into a hipGraph over fixed input blocks; per
  slice only the sliced inputs are refreshed in place and the graph is replayed.

  `partials_out` (a list; verification runs only): every slice's partial result is appended to it as a host
  float64 array before it is added (one blocking read per slice; forces the eager path).

  `reuse` (default: automatically, when `use_graph` is left alone and it saves at least a fifth of this rank's
  multiply-adds): every step runs
  once per distinct value of the cut bonds it DEPENDS on instead of once per slice (`_StagePlan`,
  `_contract_slices_staged`); the slices are then dealt to the ranks in contiguous blocks of the order in which the
  expensive steps change least.  Otherwise (`hoist_invariant`, default on) only the steps that depend on no cut bond
  are contracted once, before the slice loop (`_hoist_invariant`), and the slices run the remaining steps of the same
  path eagerly or as a hipGraph.  Every slice partial is the same tensor in all three modes.  `stats` (a dict)
  receives ``mode`` and the step / multiply-add counts."""
  comm = comm or LocalComm()
  nodes = list(nodes)
  cut_edges = list(cut_edges)
  for e in cut_edges:
    if e.is_dangling():
      raise ValueError("only contracted (non-dangling) edges can be sliced")
  be = nodes[0].backend
  dims = [e.dimension for e in cut_edges]
  all_slices = list(itertools.product(*[range(d) for d in dims])) if cut_edges else [()]

  # path on the sliced topology (cut bonds have dimension 1), searched once
  inputs, output, sizes = _index_problem(nodes)
  sliced_sizes = dict(sizes)
  for e in cut_edges:
    sliced_sizes[e] = 1
  path = algorithm(inputs, output, sliced_sizes)

  # ---- staged reuse: steps run once per distinct value of the cuts they depend on
  if reuse is None and use_graph is not None:
    reuse = False            # the caller picked the slice-by-slice machinery (eager or hipGraph) explicitly
  if cut_edges and len(all_slices) > 1 and reuse is not False and \
      not any(e.is_trace() for n in nodes for e in n.edges if not e.is_dangling()):
    plan = _StagePlan(nodes, cut_edges, path)
    blocks = plan.partition(all_slices, comm.world)          # the same on every rank (host arithmetic on the plan)
    mine = blocks[comm.rank]
    # The MODE is decided from all ranks' blocks, never from this rank's own: a rank that chose differently would
    # take a different share of the slices (some summed twice, others never -- ADVICE r4).
    with_reuse = max(plan.macs_with_reuse(b) for b in blocks)
    alone = plan.macs_alone() * max(len(b) for b in blocks)
    if reuse or with_reuse <= 0.8 * alone:
      if stats is not None:
        stats.update({"mode": "staged", "hoisted_steps": 0, "steps_per_slice": len(plan.steps),
                      "macs_alone": plan.macs_alone() * len(mine), "slices": len(mine),
                      "model_macs_with_reuse": plan.macs_with_reuse(mine), "executed_macs": 0.0})
      if mine:
        total, narrow = _contract_slices_staged(be, nodes, plan, mine, output_edge_order, partials_out, stats)
        return _finish(be, comm, total, narrow)
      # this rank got no slice: zeros of the result's shape and dtype (one slice's result times zero)
      part, narrow = _contract_slices_staged(be, nodes, plan, plan.ordered(all_slices)[:1], output_edge_order, None, None)
      return _finish(be, comm, be.multiply(part, 0.0), narrow)

  hoisted = 0
  nodes_before_hoist, cuts_before_hoist = nodes, cut_edges
  if hoist_invariant and cut_edges and len(all_slices) > 1:
    reduced = _hoist_invariant(nodes, cut_edges, path, output_edge_order)
    if reduced is not None:
      nodes, cut_edges, path, output_edge_order, hoisted = reduced
  if stats is not None:
    stats["mode"] = "slice by slice"
    stats["hoisted_steps"] = hoisted
    stats["steps_per_slice"] = sum(1 for pair in path if len(pair) > 1)

  mine = all_slices[comm.rank::comm.world]
  if stats is not None:
    inputs_r, output_r, sizes_r = _index_problem(nodes)
    for e in cut_edges:
      sizes_r[e] = 1
    per_slice = float(pathfinder.path_cost(inputs_r, output_r, sizes_r, path)[0])
    inv = 0.0
    if hoisted:
      inv = float(slicing_report(nodes_before_hoist, cuts_before_hoist, algorithm)["flops_invariant_per_slice"])
    stats.update({"slices": len(mine), "executed_macs": per_slice * max(len(mine), 1) + inv})
  if partials_out is not None:
    use_graph = False
  if use_graph is None:
    use_graph = hasattr(be, "capture") and len(cut_edges) > 0 and len(mine) >= 4
  if use_graph and mine:
    try:
      total, narrow = _contract_slices_graph(be, nodes, cut_edges, mine, path, output_edge_order)
      return _finish(be, comm, total, narrow)
    except MemoryError:
      pass   # not enough HBM for the captured sequence's fixed blocks: run the slices eagerly

  total, narrow = None, None
  for idx in mine:
    node_map, edge_map = network.copy(nodes)
    for e, i in zip(cut_edges, idx):
      network.slice_edge(edge_map[e], i, 1)
    order = [edge_map[e] for e in output_edge_order] if output_edge_order is not None else None
    part, narrow = _widen(be, contractors.contract_path(path, [node_map[n] for n in nodes], order).tensor)
    if partials_out is not None:
      partials_out.append(np.asarray(part, dtype=np.float64).copy())
    total = part if total is None else be.addition(total, part)
    for n in node_map.values():     # this slice's copies are ours: drop their tensors now (Node <-> Edge
      n.tensor = None               # cycles would otherwise keep the HBM until the cyclic GC runs)
      n.edges = []
  if total is None:
    # this rank got no slice: contribute zeros of the right shape/dtype
    node_map, edge_map = network.copy(nodes)
    for e in cut_edges:
      network.slice_edge(edge_map[e], 0, 1)
    order = [edge_map[e] for e in output_edge_order] if output_edge_order is not None else None
    part, narrow = _widen(be, contractors.contract_path(path, [node_map[n] for n in nodes], order).tensor)
    total = be.multiply(part, 0.0)
  return _finish(be, comm, total, narrow)


def _is_half(tensor):
  return getattr(tensor, "dtype", None) is not None and str(tensor.dtype) in ("bfloat16", "float16")


def _widen(be, tensor):
  """Slice partials of a bf16 / f16 network are ADDED in fp32: hundreds of slices summed in an
  8-bit mantissa would lose the small ones (the per-slice GEMMs already accumulate in fp32).
  Returns (tensor to accumulate, the half dtype it was widened from or None)."""
  if _is_half(tensor) and hasattr(be, "cast"):
    return be.cast(tensor, np.float32), tensor.dtype
  return tensor, None


def _finish(be, comm, total, narrow):
  """ONE all-reduce of the (fp32-accumulated) partial sums, then a single rounding back to the dtype the
  per-slice contraction itself produced (`narrow`; None = nothing was widened, e.g. an f32 / f64 network, a
  backend with half_output="float32", or a mixed network whose result is already wide) -- the sliced result
  has the dtype of the unsliced contraction of the same network."""
  out = comm.all_reduce_sum(be, total)
  if narrow is not None and not _is_half(out):
    out = be.cast(out, narrow)
  return out


def _contract_slices_graph(be, nodes, cut_edges, slices, path, output_edge_order):
  """hipGraph replay of one contraction path over many slices (hip backend only)."""
  # fixed input blocks: the tensors of slice `slices[0]`
  node_map, edge_map = network.copy(nodes)
  for e, i in zip(cut_edges, slices[0]):
    network.slice_edge(edge_map[e], i, 1)
  staged = [node_map[n].tensor for n in nodes]
  # which inputs depend on the slice index, and how: node position -> [(axis, cut number)]
  pos = {id(n): k for k, n in enumerate(nodes)}
  windows: Dict[int, List[Tuple[int, int]]] = {}
  for c, e in enumerate(cut_edges):
    for nd, ax in e.ends():
      windows.setdefault(pos[id(nd)], []).append((ax, c))

  def run(*tensors):
    node_map2, edge_map2 = network.copy(nodes)       # topology only; tensors are replaced below
    for n, t in zip(nodes, tensors):
      node_map2[n].tensor = t
    order = [edge_map2[e] for e in output_edge_order] if output_edge_order is not None else None
    return contractors.contract_path(path, [node_map2[n] for n in nodes], order).tensor

  graph = be.capture(run, *staged)
  try:
    total, narrow = None, None
    for idx in slices:
      for k, wins in windows.items():
        starts = [0] * nodes[k].tensor.ndim
        for ax, c in wins:
          starts[ax] = idx[c]
        be.slice_into(staged[k], nodes[k].tensor, starts)
      part, narrow = _widen(be, graph.launch())
      total = be.multiply(part, 1.0) if total is None else be.addition(total, part)
  finally:
    be.synchronize()
    graph.close()
  return total, narrow


def slicing_report(nodes: Sequence[network.Node], cut_edges: Sequence[network.Edge],
                   algorithm: Callable = pathfinder.greedy, world: int = 1) -> Dict[str, float]:
  """Cost model of a slicing plan (host only): flops and peak intermediate, sliced vs unsliced; what `world` ranks
  execute when every step runs once per value of the cuts it depends on (contract_sliced's default mode)."""
  inputs, output, sizes = _index_problem(nodes)
  path0 = algorithm(inputs, output, sizes)
  flops0, peak0 = pathfinder.path_cost(inputs, output, sizes, path0)
  sliced = dict(sizes)
  n_slices = 1
  for e in cut_edges:
    n_slices *= sizes[e]
    sliced[e] = 1
  path1 = algorithm(inputs, output, sliced)
  flops1, peak1 = pathfinder.path_cost(inputs, output, sliced, path1)
  # the part of a slice that does not depend on the cut bonds (contract_sliced runs it once: _hoist_invariant)
  nodes = list(nodes)
  touched = {id(nd) for e in cut_edges for nd, _ in e.ends()}
  steps = _variant_steps(len(nodes), [id(n) in touched for n in nodes], path1)
  invariant, remaining = 0, [frozenset(x) for x in inputs]
  live = list(range(len(nodes)))
  for ia, ib, new, is_variant in steps:
    k1, k2 = remaining[live.index(ia)], remaining[live.index(ib)]
    others = set(output)
    for x, k in zip(live, remaining):
      if x not in (ia, ib):
        others |= k
    if not is_variant:
      invariant += pathfinder._size(k1 | k2, sliced)      # pylint: disable=protected-access
    keep = [(x, k) for x, k in zip(live, remaining) if x not in (ia, ib)]
    live = [x for x, _ in keep] + [new]
    remaining = [k for _, k in keep] + [frozenset(d for d in (k1 | k2) if d in others)]
  reuse = {}
  if cut_edges and n_slices > 1 and not any(e.is_trace() for n in nodes for e in n.edges if not e.is_dangling()):
    plan = _StagePlan(nodes, list(cut_edges), path1)
    blocks = plan.partition(list(itertools.product(*[range(e.dimension) for e in cut_edges])), max(int(world), 1))
    per_rank = [plan.macs_with_reuse(b) for b in blocks]
    reuse = {"flops_with_reuse_all_ranks": float(sum(per_rank)), "flops_with_reuse_slowest_rank": float(max(per_rank)),
             "model_seconds_slowest_rank": float(max(plan.seconds_with_reuse(b) for b in blocks)),
             "slices_per_rank": [len(b) for b in blocks],
             "reuse_classes": {",".join(str(k) for k in sorted(c)) or "-": float(m) for c, m in plan.class_macs.items()},
             "staged_by_default": bool(max(per_rank) <= 0.8 * plan.macs_alone() * max(len(b) for b in blocks))}
  return {"n_slices": n_slices, "flops_unsliced": float(flops0), "peak_unsliced": float(peak0),
          "flops_per_slice": float(flops1), "peak_per_slice": float(peak1),
          "flops_invariant_per_slice": float(invariant), **reuse,
          "invariant_steps": sum(1 for st in steps if not st[3]), "steps_per_slice": len(steps),
          "overhead": float(flops1) * n_slices / max(float(flops0), 1.0)}


# ------------------------------------------------------- one very large pairwise contraction
def shard_rows(n_rows: int, world: int) -> List[Tuple[int, int]]:
  """[start, stop) of every rank's block of a leading axis of length n_rows (balanced, contiguous)."""
  base, extra = divmod(int(n_rows), int(world))
  out, start = [], 0
  for r in range(world):
    stop = start + base + (1 if r < extra else 0)
    out.append((start, stop))
    start = stop
  return out


def tensordot_sharded(backend, a, b, axes, comm=None, a_is_local=False, gather=True):
  """One big ``tensordot`` spread over the ranks by the FIRST free axis of ``a`` (the M side of
  the flattened GEMM): rank r contracts rows [start_r, stop_r) of ``a`` with all of ``b`` on its
  own GPU -- no exchange during the contraction -- and the row blocks of the result are
  concatenated by ONE all-gather (RCCL over xGMI) only if the consumer needs the whole tensor
  (``gather=True``); otherwise every rank keeps its block.  K is never sharded: a K split would
  all-reduce the full M x N fp32 result (17 GB at D = 256) over 153 GB/s links.

  ``a`` is either the full operand (replicated; each rank slices its block) or, with
  ``a_is_local=True``, already this rank's block along that axis.  Requires the first axis of
  ``a`` to be a free (uncontracted) axis.  Returns ``(tensor, (start, stop))``."""
  comm = comm or LocalComm()
  shape_a = tuple(backend.shape_tuple(a))
  try:
    iter(axes)
    axes_a = axes[0]
    axes_a = [int(x) for x in axes_a] if hasattr(axes_a, "__iter__") else [int(axes_a)]
  except TypeError:
    axes_a = list(range(len(shape_a) - int(axes), len(shape_a)))
  if 0 in [x % max(len(shape_a), 1) for x in axes_a]:
    raise ValueError("tensordot_sharded shards the first axis of `a`, which must not be contracted")
  if a_is_local:
    counts = comm.all_gather_counts(shape_a[0]) if hasattr(comm, "all_gather_counts") else [shape_a[0]]
    start = sum(counts[:comm.rank])
    bounds = (start, start + shape_a[0])
    local = a
  else:
    blocks = shard_rows(shape_a[0], comm.world)
    counts = [e - s for s, e in blocks]
    bounds = blocks[comm.rank]
    local = backend.slice(a, (bounds[0],) + (0,) * (len(shape_a) - 1),
                          (bounds[1] - bounds[0],) + shape_a[1:])
  part = backend.tensordot(local, b, axes)
  if not gather or comm.world == 1:
    return part, bounds
  return comm.all_gather_rows(backend, part, counts), (0, sum(counts))
