"""Contraction-path search with the ``opt_einsum.paths`` calling convention.

The reference's contractors hand the search to the third-party ``opt_einsum``
package (``contractors/opt_einsum_paths/path_contractors.py:124-125, 160-161,
192``; protocol in ``utils.py:29-46``)::

    path = algorithm(input_sets, output_set, size_dict, memory_limit=None)

``input_sets`` is a list of sets of hashable index labels (the reference passes
``Edge`` objects), ``path`` a list of index pairs into the *current* operand
list (contracted operands removed, result appended).  ``opt_einsum`` is not
installed here (requirements.txt:3, un-vendored), so its published algorithms
are restated from their description: greedy = repeatedly contract the pair with
the most negative ``size(out) - size(a) - size(b)``, outer products last,
smallest first; optimal = exact minimum-flop search; branch = depth-first search
over the ``nbranch`` best greedy candidates per step.  Host-only integer work.
"""
import heapq
import math
import itertools

import numpy as np
from typing import Dict, Hashable, List, Optional, Sequence, Set, Tuple

Path = List[Tuple[int, int]]


def _size(indices, sizes) -> int:
  out = 1
  for i in indices:
    out *= int(sizes[i])
  return out


def ssa_to_linear(ssa_path: Sequence[Tuple[int, ...]], num_inputs: int) -> Path:
  """Static-single-assignment ids -> positions in the shrinking operand list.

  Inputs carry ids 0..num_inputs-1; every contraction creates the next id."""
  ids = list(range(num_inputs))
  nxt = num_inputs
  path = []
  for pair in ssa_path:
    pos = tuple(sorted(ids.index(i) for i in pair))
    path.append(pos)
    for p in sorted(pos, reverse=True):
      del ids[p]
    ids.append(nxt)
    nxt += 1
  return path


def _result_indices(k1, k2, output, ref_counts_get):
  """Indices surviving the contraction of operands k1 and k2."""
  either = k1 | k2
  two = k1 & k2
  one = either - two
  keep = set()
  for d in either:
    if d in output:
      keep.add(d)
    elif d in two:
      if ref_counts_get(d) > 2:
        keep.add(d)
    elif d in one:
      if ref_counts_get(d) > 1:
        keep.add(d)
  return frozenset(keep)


def greedy(inputs: List[Set[Hashable]], output: Set[Hashable], size_dict: Dict[Hashable, int],
           memory_limit: Optional[int] = None) -> Path:  # pylint: disable=unused-argument
  """Greedy pairwise contraction order (cost = bytes removed by the step)."""
  n = len(inputs)
  if n == 0:
    return []
  if n == 1:
    return [(0,)]
  output = frozenset(output)
  keys = [frozenset(s) for s in inputs]
  ssa_path: List[Tuple[int, ...]] = []
  next_id = n

  # operands with identical index sets are multiplied together first (Hadamard)
  remaining: Dict[frozenset, int] = {}
  for ssa_id, key in enumerate(keys):
    if key in remaining:
      ssa_path.append((remaining[key], ssa_id))
      remaining[key] = next_id
      next_id += 1
    else:
      remaining[key] = ssa_id

  dim_to_keys: Dict[Hashable, Set[frozenset]] = {}
  for key in remaining:
    for d in key - output:
      dim_to_keys.setdefault(d, set()).add(key)

  def refs(d):
    return len(dim_to_keys.get(d, ())) + (1 if d in output else 0)

  footprint = {key: _size(key, size_dict) for key in remaining}
  queue: list = []

  def push(k1, k2s, push_all):
    cands = []
    for k2 in k2s:
      k12 = _result_indices(k1, k2, output, refs)
      cost = _size(k12, size_dict) - footprint[k1] - footprint[k2]
      id1, id2 = remaining[k1], remaining[k2]
      if id1 > id2:
        k1_, k2_, id1, id2 = k2, k1, id2, id1
      else:
        k1_, k2_ = k1, k2
      cands.append((cost, id1, id2, k1_, k2_, k12))
    if not cands:
      return
    if push_all:
      for c in cands:
        heapq.heappush(queue, c)
    else:
      heapq.heappush(queue, min(cands, key=lambda c: c[:3]))

  for d_keys in list(dim_to_keys.values()):
    ordered = sorted(d_keys, key=remaining.__getitem__)
    for i, k1 in enumerate(ordered[:-1]):
      push(k1, ordered[i + 1:], True)

  while queue:
    cost, id1, id2, k1, k2, k12 = heapq.heappop(queue)
    if k1 not in remaining or k2 not in remaining:
      continue  # stale candidate
    if remaining[k1] != id1 and remaining[k1] != id2:
      continue
    ssa1, ssa2 = remaining.pop(k1), remaining.pop(k2)
    for d in k1 - output:
      dim_to_keys[d].discard(k1)
    for d in k2 - output:
      dim_to_keys[d].discard(k2)
    ssa_path.append((ssa1, ssa2))
    if k12 in remaining:
      ssa_path.append((remaining[k12], next_id))
      next_id += 1
    else:
      for d in k12 - output:
        dim_to_keys.setdefault(d, set()).add(k12)
    remaining[k12] = next_id
    next_id += 1
    footprint[k12] = _size(k12, size_dict)
    k2s = {k for d in k12 - output for k in dim_to_keys.get(d, ())}
    k2s.discard(k12)
    if k2s:
      push(k12, sorted(k2s, key=remaining.__getitem__), False)

  # disconnected pieces: outer products, smallest operands first
  heap = [(_size(key & output, size_dict), ssa_id, key) for key, ssa_id in remaining.items()]
  heapq.heapify(heap)
  while len(heap) > 1:
    _, id1, k1 = heapq.heappop(heap)
    _, id2, k2 = heapq.heappop(heap)
    ssa_path.append((min(id1, id2), max(id1, id2)))
    k12 = (k1 | k2) & output
    heapq.heappush(heap, (_size(k12, size_dict), next_id, k12))
    next_id += 1
  return ssa_to_linear(ssa_path, n)


def _pair_result(k1, k2, others_and_output):
  """(surviving indices, multiply count) of contracting k1 with k2."""
  either = k1 | k2
  return frozenset(d for d in either if d in others_and_output), either


def optimal(inputs: List[Set[Hashable]], output: Set[Hashable], size_dict: Dict[Hashable, int],
            memory_limit: Optional[int] = None) -> Path:
  """Exact minimum-multiplication order.

  Dynamic programming over connected sub-networks (disconnected pieces are
  joined by outer products at the end); exhaustive enough to be exact for the
  network sizes the reference's ``optimal`` contractor is meant for.
  """
  n = len(inputs)
  if n == 0:
    return []
  if n == 1:
    return [(0,)]
  output = frozenset(output)
  keys = [frozenset(s) for s in inputs]

  def legs(mask):
    """Indices of the contracted sub-network `mask` that stay open."""
    inside, outside = {}, set(output)
    for i in range(n):
      if mask >> i & 1:
        for d in keys[i]:
          inside[d] = inside.get(d, 0) + 1
      else:
        outside |= keys[i]
    return frozenset(d for d in inside if d in outside)

  # best[mask] = (cost, tree) ; tree is an int (leaf) or a pair of trees
  best = {1 << i: (0, i) for i in range(n)}
  by_size = {1: [1 << i for i in range(n)]}
  leg_cache = {1 << i: legs(1 << i) for i in range(n)}
  for size in range(2, n + 1):
    found = {}
    for m in range(1, size // 2 + 1):
      for s1 in by_size.get(m, ()):
        for s2 in by_size.get(size - m, ()):
          if s1 & s2:
            continue
          if m == size - m and s1 > s2:
            continue
          l1, l2 = leg_cache[s1], leg_cache[s2]
          if not l1 & l2:
            continue  # no shared index: outer product, handled at the end
          mask = s1 | s2
          cost = best[s1][0] + best[s2][0] + _size(l1 | l2, size_dict)
          if memory_limit is not None:
            if mask not in leg_cache:
              leg_cache[mask] = legs(mask)
            if _size(leg_cache[mask], size_dict) > memory_limit:
              continue
          cur = found.get(mask)
          if cur is None or cost < cur[0]:
            found[mask] = (cost, (best[s1][1], best[s2][1]))
    for mask, (cost, tree) in found.items():
      best[mask] = (cost, tree)
      if mask not in leg_cache:
        leg_cache[mask] = legs(mask)
    by_size[size] = list(found)

  # connected components of the index graph, joined by outer products
  comps, seen = [], 0
  for i in range(n):
    if seen >> i & 1:
      continue
    comp, frontier = 1 << i, [i]
    while frontier:
      j = frontier.pop()
      for t in range(n):
        if not comp >> t & 1 and keys[j] & keys[t]:
          comp |= 1 << t
          frontier.append(t)
    seen |= comp
    comps.append(comp)
  if any(c not in best for c in comps):
    # memory_limit pruned every order of some component: fall back to greedy
    return greedy(inputs, output, size_dict)
  trees = sorted(((_size(leg_cache[c], size_dict), best[c][1]) for c in comps),
                 key=lambda t: t[0])
  tree = trees[0][1]
  for _, t in trees[1:]:
    tree = (tree, t)
  return _tree_to_path(tree, n)


def _tree_to_path(tree, n) -> Path:
  """Contraction tree -> linear path; deeper / later leaves are contracted first
  and each pair is emitted with the convention of the reference's golden paths."""
  ssa_path = []
  counter = [n]

  def visit(t):
    if isinstance(t, int):
      return t
    a, b = visit(t[0]), visit(t[1])
    ssa_path.append((min(a, b), max(a, b)))
    counter[0] += 1
    return counter[0] - 1

  # visit the subtree containing the highest leaf first (matches opt_einsum's
  # ordering on chains, e.g. [(2, 3), (1, 2), (0, 1)] for a 4-matrix chain)
  def max_leaf(t):
    return t if isinstance(t, int) else max(max_leaf(t[0]), max_leaf(t[1]))

  def ordered(t):
    if isinstance(t, int):
      return t
    l, r = ordered(t[0]), ordered(t[1])
    return (l, r) if max_leaf(l) > max_leaf(r) else (r, l)

  visit(ordered(tree))
  return ssa_to_linear(ssa_path, n)


def dynamic_programming(inputs, output, size_dict, memory_limit=None) -> Path:
  """Name used by ``path_contractors.optimal`` (path_contractors.py:124-125)."""
  return optimal(inputs, output, size_dict, memory_limit)


def branch(inputs: List[Set[Hashable]], output: Set[Hashable], size_dict: Dict[Hashable, int],
           memory_limit: Optional[int] = None, nbranch: Optional[int] = None) -> Path:
  """Depth-first search over the ``nbranch`` best greedy candidates per step.

  ``nbranch=None`` explores every connected pair (exact); ``nbranch=1`` is a
  greedy descent.  Total multiplication count decides between finished paths.
  """
  n = len(inputs)
  if n == 0:
    return []
  if n == 1:
    return [(0,)]
  output = frozenset(output)
  start = [frozenset(s) for s in inputs]
  best = {"flops": None, "path": None}

  def recurse(path, remaining, flops):
    if len(remaining) == 1:
      if best["flops"] is None or flops < best["flops"]:
        best["flops"], best["path"] = flops, path
      return
    cands = []
    for i, j in itertools.combinations(range(len(remaining)), 2):
      k1, k2 = remaining[i], remaining[j]
      if not k1 & k2:
        continue
      others = set(output)
      for t, k in enumerate(remaining):
        if t not in (i, j):
          others |= k
      k12, either = _pair_result(k1, k2, others)
      size12 = _size(k12, size_dict)
      if memory_limit is not None and size12 > memory_limit:
        continue
      new_flops = flops + _size(either, size_dict)
      if best["flops"] is not None and new_flops >= best["flops"]:
        continue
      cost = size12 - _size(k1, size_dict) - _size(k2, size_dict)
      cands.append((cost, new_flops, i, j, k12))
    if not cands:
      # only outer products left (or everything pruned): join the two smallest
      order = sorted(range(len(remaining)), key=lambda t: _size(remaining[t], size_dict))
      i, j = sorted(order[:2])
      k12 = (remaining[i] | remaining[j])
      others = set(output)
      for t, k in enumerate(remaining):
        if t not in (i, j):
          others |= k
      k12 = frozenset(d for d in k12 if d in others)
      new_flops = flops + _size(remaining[i] | remaining[j], size_dict)
      if best["flops"] is not None and new_flops >= best["flops"]:
        return
      cands = [(0, new_flops, i, j, k12)]
    cands.sort(key=lambda c: c[:4])
    if nbranch is not None:
      cands = cands[:nbranch]
    for _, new_flops, i, j, k12 in cands:
      nxt = [k for t, k in enumerate(remaining) if t not in (i, j)] + [k12]
      recurse(path + [(i, j)], nxt, new_flops)

  recurse([], start, 0)
  if best["path"] is None:
    return greedy(inputs, output, size_dict)
  return best["path"]


def auto(inputs, output, size_dict, memory_limit=None) -> Path:
  """Size-based choice of ``path_contractors.auto`` (path_contractors.py:197-265)."""
  n = len(inputs)
  if n < 5:
    return optimal(inputs, output, size_dict, memory_limit)
  if n < 7:
    return branch(inputs, output, size_dict, memory_limit, nbranch=None)
  if n < 9:
    return branch(inputs, output, size_dict, memory_limit, nbranch=2)
  if n < 15:
    return branch(inputs, output, size_dict, memory_limit, nbranch=1)
  return greedy(inputs, output, size_dict, memory_limit)


def path_cost(inputs, output, size_dict, path) -> Tuple[int, int]:
  """(total multiply-adds, largest intermediate size) of a linear path."""
  remaining = [frozenset(s) for s in inputs]
  output = frozenset(output)
  flops, peak = 0, 0
  for pair in path:
    if len(pair) == 1:
      continue
    i, j = sorted(pair)
    k1, k2 = remaining[i], remaining[j]
    others = set(output)
    for t, k in enumerate(remaining):
      if t not in (i, j):
        others |= k
    k12 = frozenset(d for d in (k1 | k2) if d in others)
    flops += _size(k1 | k2, size_dict)
    peak = max(peak, _size(k12, size_dict))
    remaining = [k for t, k in enumerate(remaining) if t not in (i, j)] + [k12]
  return flops, peak


def path_depth(path, num_inputs: int) -> int:
  """Depth of the contraction tree of a linear path: the largest number of pairwise contractions between an input
  and the result.  Every intermediate of a bf16 / f16 contraction is rounded once, so a product term meets at most
  `depth` roundings -- the a-priori error bound ((1 + 2^-9)^depth - 1) * sum |terms| used by bench.py's checks."""
  depth = [0] * num_inputs
  for pair in path:
    if len(pair) == 1:
      continue
    i, j = sorted(pair)
    d = max(depth[i], depth[j]) + 1
    depth = [x for t, x in enumerate(depth) if t not in (i, j)] + [d]
  return max(depth) if depth else 0


# ----------------------------------------------------------------- ncon-style networks
def _ncon_sets(tensors, labels):
  """ncon label lists -> (index sets, output set, sizes); a label that repeats on one tensor is a partial
  trace and costs nothing in the pairwise model, so it is dropped from that tensor's set."""
  sizes, inputs = {}, []
  for t, labs in zip(tensors, labels):
    labs = [int(l) for l in labs]
    if len(labs) != len(t.shape):
      raise ValueError(f"labels {labs} do not match a tensor of shape {tuple(t.shape)}")
    for l, d in zip(labs, t.shape):
      sizes[l] = int(d)
    inputs.append({l for l in labs if labs.count(l) == 1})
  output = {l for s in inputs for l in s if l < 0}
  return inputs, output, sizes


def _ncon_order_from_path(inputs, path):
  """Positive labels in the order a linear pairwise path closes them (the `con_order` of ncon)."""
  remaining = [set(s) for s in inputs]
  order = []
  for pair in path:
    if len(pair) == 1:
      continue
    i, j = sorted(pair)
    k1, k2 = remaining[i], remaining[j]
    closed = sorted(l for l in (k1 & k2) if l > 0)
    order.extend(closed)
    merged = (k1 | k2) - set(closed)
    remaining = [k for t, k in enumerate(remaining) if t not in (i, j)] + [merged]
  return order


def ncon_cost_check(tensors, labels, con_order) -> float:
  """log10 of the multiply count of contracting an ncon network in the given label order, pairwise, all
  labels shared by a pair closing together (nconinterface.py:124-180)."""
  inputs, _, sizes = _ncon_sets(tensors, labels)
  remaining = [set(s) for s in inputs]
  total = 0
  todo = [int(l) for l in con_order]
  while todo:
    lab = todo[0]
    holders = [t for t, k in enumerate(remaining) if lab in k]
    if len(holders) != 2:                 # traced on one tensor (already dropped) or closed with a partner
      todo.pop(0)
      continue
    i, j = holders
    closed = {l for l in remaining[i] & remaining[j] if l > 0}
    total += _size(remaining[i] | remaining[j], sizes)
    merged = (remaining[i] | remaining[j]) - closed
    remaining = [k for t, k in enumerate(remaining) if t not in (i, j)] + [merged]
    todo = [l for l in todo if l not in closed]
  while len(remaining) > 1:               # disconnected pieces: outer products
    a, b = remaining.pop(), remaining.pop()
    total += _size(a | b, sizes)
    remaining.append(a | b)
  return math.log10(total) if total > 0 else 0.0


def ncon_solver(tensors, labels, max_branch: Optional[int] = None):
  """Cheapest contraction order of an ncon-style network (nconinterface.py:21-45).  Returns
  `(con_order, log10(multiply count), is_optimal)`: `max_branch=None` runs the exact dynamic programme,
  otherwise a branch search over the `max_branch` best candidates per step (`max_branch=1`: greedy)."""
  inputs, output, sizes = _ncon_sets(tensors, labels)
  if max_branch is None:
    path, exact = optimal(inputs, output, sizes), True
  else:
    path, exact = branch(inputs, output, sizes, nbranch=int(max_branch)), False
  macs, _ = path_cost(inputs, output, sizes, path)
  order = _ncon_order_from_path(inputs, path)
  traced = sorted({int(l) for labs in labels for l in labs if list(labs).count(l) == 2})
  con_order = np.array(traced + order, dtype=int)
  return con_order, (math.log10(macs) if macs > 0 else 0.0), exact
