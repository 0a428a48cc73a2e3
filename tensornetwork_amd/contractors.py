"""Path-driven contractors (``greedy``, ``optimal``, ``branch``, ``auto``, ``custom``).

Mirror of ``tensornetwork/contractors/opt_einsum_paths/path_contractors.py``:
``base`` (36-97) contracts trace edges, asks a path algorithm for the pairwise
order over ``(input_sets, output_set, size_dict)`` built from the nodes' edges
(``utils.get_path``, utils.py:29-46), then runs one ``contract_between`` per
step.  The path algorithms come from ``tensornetwork_amd.pathfinder`` (the
reference delegates to the absent third-party ``opt_einsum``).

Works on ``tensornetwork_amd.network.Node`` objects; the search itself is pure
host code, the pairwise steps are GEMMs on the backend.
"""
import functools
from typing import Any, Dict, Callable, Iterable, List, Optional, Sequence, Set, Tuple

from tensornetwork_amd import network, pathfinder


# Path cache: the search depends only on the index structure (which node carries which
# bond, bond sizes, which bonds stay open) and on the algorithm + its keyword arguments,
# never on tensor values.  Iterative workloads (MERA sweeps, sliced networks, DMRG) ask
# for the same topology thousands of times; the search costs ms, a small contraction us.
_PATH_CACHE: dict = {}
_PATH_CACHE_MAX = 256


def _topology_key(input_sets, output_set, size_dict, algorithm):
  label = {}
  inputs = []
  for node_edges in input_sets:   # edges relabelled by first appearance, in axis order
    inputs.append(tuple(label.setdefault(e, len(label)) for e in node_edges))
  sizes = tuple(size_dict[e] for e in label)
  out = tuple(sorted(label[e] for e in output_set))
  if isinstance(algorithm, functools.partial):
    alg = (getattr(algorithm.func, "__qualname__", repr(algorithm.func)), id(algorithm.func),
           algorithm.args, tuple(sorted(algorithm.keywords.items())))
  else:
    alg = (getattr(algorithm, "__qualname__", repr(algorithm)), id(algorithm))
  return (tuple(inputs), sizes, out, alg)


def get_path(nodes: Iterable[network.Node], algorithm: Callable) -> Tuple[List[Tuple[int, int]], List[network.Node]]:
  nodes = list(nodes)
  input_lists = [list(node.edges) for node in nodes]
  input_sets = [set(edges) for edges in input_lists]
  output_set = network.get_subgraph_dangling(nodes)
  size_dict = {edge: edge.dimension for edge in network.get_all_edges(nodes)}
  try:
    key = _topology_key(input_lists, output_set, size_dict, algorithm)
    hash(key)
  except TypeError:
    key = None
  if key is not None and key in _PATH_CACHE:
    return list(_PATH_CACHE[key]), nodes
  path = algorithm(input_sets, output_set, size_dict)
  if key is not None:
    if len(_PATH_CACHE) >= _PATH_CACHE_MAX:
      _PATH_CACHE.pop(next(iter(_PATH_CACHE)))
    _PATH_CACHE[key] = [tuple(p) for p in path]
  return path, nodes


def _edge_times(path: Sequence[Tuple[int, ...]], nodes: Sequence[network.Node]) -> Dict[network.Edge, int]:
  """edge -> index of the path step that contracts it (host-only dry run of the path over sets of
  original nodes).  Feeds the layout planning of network.contract_between."""
  groups = [{id(n)} for n in nodes]
  owner_edges = [set(n.edges) for n in nodes]
  times: Dict[network.Edge, int] = {}
  for step, pair in enumerate(path):
    if len(pair) == 1:
      continue
    a, b = sorted(pair)
    for e in owner_edges[a] & owner_edges[b]:
      times[e] = step
    merged_nodes = groups[a] | groups[b]
    merged_edges = owner_edges[a] ^ owner_edges[b]
    groups = [g for i, g in enumerate(groups) if i not in (a, b)] + [merged_nodes]
    owner_edges = [g for i, g in enumerate(owner_edges) if i not in (a, b)] + [merged_edges]
  return times


def contract_labelled(be, operands: Dict[int, Tuple[Any, Sequence[Any]]], steps: Sequence[Tuple[int, int, int]],
                      label_time: Optional[Dict[Any, int]] = None) -> Dict[int, Tuple[Any, List[Any]]]:
  """Pairwise contractions on operands given as {id: (tensor, axis labels)}: axes that carry the same label are
  connected, a label that occurs once stays open (its partner lives elsewhere -- another stage of a staged
  contraction -- or it is a dangling leg).  `steps` = [(id_a, id_b, id_new)] in execution order.  Returns
  {id: (tensor, axis labels)} of what is left.  Layout planning as in `contract_path`: `label_time` says at which step
  of the whole path a label is contracted (network.contract_between's `edge_time`)."""
  nodes = {k: network.Node(t, backend=be) for k, (t, _) in operands.items()}
  edge_label, first = {}, {}
  for k, (_, labels) in operands.items():
    for ax, lab in enumerate(labels):
      if lab in first:
        k0, ax0 = first.pop(lab)
        edge_label[network.connect(nodes[k0][ax0], nodes[k][ax])] = lab
      else:
        first[lab] = (k, ax)
  for lab, (k, ax) in first.items():
    edge_label[nodes[k][ax]] = lab
  edge_time = None
  if label_time is not None:
    edge_time = {e: label_time[lab] for e, lab in edge_label.items() if lab in label_time}
  for ia, ib, new in steps:
    a, b = nodes.pop(ia), nodes.pop(ib)
    nodes[new] = network.contract_between(a, b, allow_outer_product=True, edge_time=edge_time)
    for used in (a, b):               # Node <-> Edge cycles: drop the references now (intermediates can be GBs; the
      used.tensor, used.edges = None, []   # operands' tensors stay alive with whoever passed them in)
  out = {k: (nd.tensor, [edge_label[e] for e in nd.edges]) for k, nd in nodes.items()}
  for nd in nodes.values():
    nd.tensor, nd.edges = None, []
  return out


def contract_path(path: Sequence[Tuple[int, ...]], nodes: Iterable[network.Node],
                  output_edge_order: Optional[Sequence[network.Edge]] = None) -> network.Node:
  """Run a linear path (path_contractors.py:354-403)."""
  nodes = list(nodes)
  if not nodes:
    raise ValueError("No node was given to contract.")
  edges = network.get_all_edges(nodes)
  for edge in edges:
    if not edge.is_dangling() and edge.is_trace():
      node = edge.node1
      if node in nodes:
        new = network.contract_trace_edges(node)
        nodes[nodes.index(node)] = new
  edge_time = _edge_times(path, nodes)
  given = {id(n) for n in nodes}
  for pair in path:
    if len(pair) == 1:
      continue
    a, b = sorted(pair)
    new = network.contract_between(nodes[a], nodes[b], allow_outer_product=True, edge_time=edge_time)
    for used in (nodes[a], nodes[b]):
      if id(used) not in given:
        # an intermediate this function created: nobody else holds it.  Node <-> Edge references are
        # cycles, so without this its tensor (possibly many GB of HBM) would wait for the cyclic GC.
        used.tensor = None
        used.edges = []
    nodes = [n for i, n in enumerate(nodes) if i not in (a, b)] + [new]
  final = nodes[0]
  if len(nodes) != 1:
    raise ValueError("the path did not reduce the network to a single node")
  if output_edge_order is not None:
    final.reorder_edges(list(output_edge_order))
  return final


def base(nodes: Iterable[network.Node], algorithm: Callable,
         output_edge_order: Optional[Sequence[network.Edge]] = None,
         ignore_edge_order: bool = False) -> network.Node:
  nodes = list(nodes)
  if not nodes:
    raise ValueError("No node was given to contract.")
  dangling = network.get_subgraph_dangling(nodes)
  if output_edge_order is None:
    output_edge_order = list(dangling)
    if len(output_edge_order) > 1 and not ignore_edge_order:
      raise ValueError("The final node after contraction has more than one remaining edge. In this "
                       "case `output_edge_order` has to be provided.")
  if set(output_edge_order) != dangling:
    raise ValueError("output edges are not equal to the remaining non-contracted edges of the final node.")
  # trace edges first (path_contractors.py:73-77)
  for i, node in enumerate(nodes):
    if any(e.is_trace() for e in node.edges):
      nodes[i] = network.contract_trace_edges(node)
  if len(nodes) == 1:
    final = nodes[0]
  else:
    path, nodes = get_path(nodes, algorithm)
    final = contract_path(path, nodes)
  if not ignore_edge_order and len(final.edges) > 1:
    final.reorder_edges(list(output_edge_order))
  return final


def greedy(nodes, output_edge_order=None, memory_limit: Optional[int] = None, ignore_edge_order: bool = False):
  """path_contractors.py:165-193."""
  alg = functools.partial(pathfinder.greedy, memory_limit=memory_limit)
  return base(nodes, alg, output_edge_order, ignore_edge_order)


def optimal(nodes, output_edge_order=None, memory_limit: Optional[int] = None, ignore_edge_order: bool = False):
  """path_contractors.py:100-126."""
  alg = functools.partial(pathfinder.optimal, memory_limit=memory_limit)
  return base(nodes, alg, output_edge_order, ignore_edge_order)


def branch(nodes, output_edge_order=None, memory_limit: Optional[int] = None, nbranch: Optional[int] = None,
           ignore_edge_order: bool = False):
  """path_contractors.py:129-162."""
  alg = functools.partial(pathfinder.branch, memory_limit=memory_limit, nbranch=nbranch)
  return base(nodes, alg, output_edge_order, ignore_edge_order)


def auto(nodes, output_edge_order=None, memory_limit: Optional[int] = None, ignore_edge_order: bool = False):
  """path_contractors.py:197-265."""
  nodes = list(nodes)
  if len(nodes) == 1 and output_edge_order is None:
    output_edge_order = list(nodes[0].get_all_dangling()) if not ignore_edge_order else None
  alg = functools.partial(pathfinder.auto, memory_limit=memory_limit)
  return base(nodes, alg, output_edge_order, ignore_edge_order)


def custom(nodes, optimizer: Callable, output_edge_order=None, memory_limit: Optional[int] = None,
           ignore_edge_order: bool = False):
  """Any callable with the opt_einsum path signature (path_contractors.py:268-296)."""
  alg = functools.partial(optimizer, memory_limit=memory_limit)
  return base(nodes, alg, output_edge_order, ignore_edge_order)


def path_solver(algorithm: str, nodes, memory_limit: Optional[int] = None, nbranch: Optional[int] = None):
  """Only compute the path (path_contractors.py:299-351)."""
  if algorithm == "optimal":
    alg = functools.partial(pathfinder.optimal, memory_limit=memory_limit)
  elif algorithm == "branch":
    alg = functools.partial(pathfinder.branch, memory_limit=memory_limit, nbranch=nbranch)
  elif algorithm == "greedy":
    alg = functools.partial(pathfinder.greedy, memory_limit=memory_limit)
  elif algorithm == "auto":
    alg = functools.partial(pathfinder.auto, memory_limit=memory_limit)
  else:
    raise ValueError("algorithm {algorithm} not implemented".format(algorithm=algorithm))
  path, _ = get_path(list(nodes), alg)
  return path


def bucket(nodes: Iterable[network.Node], contraction_order: Sequence["network.CopyNode"]) -> Set[network.Node]:
  """Bucket elimination over copy tensors (bucket_contractor.py:21-57, arXiv:1712.05384): each copy node
  in `contraction_order` is contracted with all of its neighbours in one hyper-index einsum
  (`network.contract_copy_node`), never forming the dense copy tensor.  Returns the remaining nodes --
  finish with another contractor if copy nodes alone do not close the network."""
  remaining = {id(n): n for n in nodes}
  for copy_node in contraction_order:
    partners = copy_node.get_partners()
    new_node = network.contract_copy_node(copy_node)
    for gone in list(partners) + [copy_node]:
      remaining.pop(id(gone), None)
    remaining[id(new_node)] = new_node
  return set(remaining.values())
