"""Path search: the 9 golden paths of the reference's
contractors/opt_einsum_paths/path_calculation_test.py:84-94 (opt_einsum's own
expected outputs), plus cost bookkeeping."""
import numpy as np
import pytest

from tensornetwork_amd import pathfinder as pf


def _network(shapes, connections):
  label, nxt = {}, 0
  for x, y in connections:
    label[x] = label[y] = nxt
    nxt += 1
  inputs, sizes, output = [], {}, set()
  for n, shape in enumerate(shapes):
    s = set()
    for ax, d in enumerate(shape):
      if (n, ax) in label:
        l = label[(n, ax)]
      else:
        l, nxt = nxt, nxt + 1
        output.add(l)
      s.add(l)
      sizes[l] = d
    inputs.append(s)
  return inputs, output, sizes


def gemm_network():  # path_calculation_test.py:37-47
  return _network([(1, 2, 4), (1, 3), (2, 4, 3)],
                  [((0, 0), (1, 0)), ((0, 1), (2, 0)), ((0, 2), (2, 1)), ((1, 1), (2, 2))])


def inner_network():  # :50-60
  return _network([(5, 2, 3, 4), (5, 3), (2, 4)],
                  [((0, 0), (1, 0)), ((0, 1), (2, 0)), ((0, 2), (1, 1)), ((0, 3), (2, 1))])


def matrix_chain():  # :63-73
  d = [10, 8, 6, 4, 2]
  return _network(list(zip(d[:-1], d[1:])), [((i, 1), (i + 1, 0)) for i in range(3)])


GOLDEN = [
    ("optimal", gemm_network, [(0, 2), (0, 1)]),
    ("branch", gemm_network, [(0, 2), (0, 1)]),
    ("greedy", gemm_network, [(0, 2), (0, 1)]),
    ("optimal", inner_network, [(0, 1), (0, 1)]),
    ("branch", inner_network, [(0, 1), (0, 1)]),
    ("greedy", inner_network, [(0, 1), (0, 1)]),
    ("optimal", matrix_chain, [(2, 3), (1, 2), (0, 1)]),
    ("branch", matrix_chain, [(2, 3), (1, 2), (0, 1)]),
    ("greedy", matrix_chain, [(0, 1), (0, 2), (0, 1)]),
]


@pytest.mark.parametrize("alg,net,expected", GOLDEN)
def test_golden_paths(alg, net, expected):
  path = getattr(pf, alg)(*net())
  assert isinstance(path, list) and all(isinstance(p, tuple) for p in path)
  assert path == expected


def test_mps_mpo_optimal_path():
  # path_contractors_node_test.py:200-217: optimal on the MPS-MPO-L network -> [(1, 3), (1, 2), (0, 1)]
  D, d, M = 100, 4, 10
  shapes = [(D, d, D), (D, M, D), (M, M, d, d), (D, d, D)]
  conns = [((0, 0), (1, 0)), ((0, 1), (2, 2)), ((1, 1), (2, 0)), ((2, 3), (3, 1)), ((1, 2), (3, 0))]
  path = pf.optimal(*_network(shapes, conns))
  assert path == [(1, 3), (1, 2), (0, 1)]


def test_paths_are_valid_and_optimal_not_worse_than_greedy():
  rng = np.random.default_rng(1)
  import networkx as nx
  for seed in range(4):
    g = nx.random_regular_graph(3, 10, seed=seed)
    idx = {e: i for i, e in enumerate(g.edges)}
    inputs = [set() for _ in g.nodes]
    for (x, y), i in idx.items():
      inputs[x].add(i)
      inputs[y].add(i)
    sizes = {i: int(rng.integers(2, 5)) for i in idx.values()}
    costs = {}
    for alg in ("greedy", "optimal", "branch", "auto"):
      path = getattr(pf, alg)(inputs, set(), sizes)
      assert len(path) == len(inputs) - 1
      n = len(inputs)
      for a, b in path:  # indices valid in the shrinking list
        assert 0 <= a < b < n
        n -= 1
      costs[alg] = pf.path_cost(inputs, set(), sizes, path)[0]
    assert costs["optimal"] <= costs["greedy"]
    assert costs["optimal"] <= costs["branch"] or costs["branch"] <= costs["greedy"]


def test_disconnected_network_uses_outer_products():
  inputs, sizes = [{0}, {0}, {1}, {1}], {0: 3, 1: 4}
  for alg in ("greedy", "optimal", "branch"):
    path = getattr(pf, alg)(inputs, set(), sizes)
    assert len(path) == 3


def test_path_depth_of_chain_and_balanced_tree():
  """path_depth = roundings a product term can meet: a left-to-right chain of 5 tensors has depth 4, a balanced
  tree of 4 tensors depth 2 (bench.py's a-priori bound for bf16 contractions uses it)."""
  from tensornetwork_amd import pathfinder
  assert pathfinder.path_depth([(0, 1), (0, 3), (0, 2), (0, 1)], 5) == 4      # result always appended, then reused
  assert pathfinder.path_depth([(0, 1), (0, 1), (0, 1)], 4) == 2             # (a b) (c d) then the two results
  assert pathfinder.path_depth([], 1) == 0
