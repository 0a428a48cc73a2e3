"""Node-level helpers of the reference's `linalg/node_linalg.py`: initialisers that return Nodes, `conj`,
`transpose`, `norm` and the operator Kronecker product `kron` (edges ordered inputs first, outputs last)."""
from typing import Any, List, Optional, Sequence

from tensornetwork_amd import network
from tensornetwork_amd.network import Node


def initialize_node(fname: str, *fargs: Any, name: Optional[str] = None, axis_names: Optional[List[str]] = None,
                    backend=None, **fkwargs: Any) -> Node:
  """Node around `backend.<fname>(*fargs, **fkwargs)` (node_linalg.py:32-64)."""
  be = network._resolve_backend(backend)  # pylint: disable=protected-access
  return Node(getattr(be, fname)(*fargs, **fkwargs), name=name, axis_names=axis_names, backend=be)


def eye(N: int, dtype=None, M: Optional[int] = None, name=None, axis_names=None, backend=None) -> Node:  # pylint: disable=invalid-name
  return initialize_node("eye", N, name=name, axis_names=axis_names, backend=backend, dtype=dtype, M=M)


def zeros(shape: Sequence[int], dtype=None, name=None, axis_names=None, backend=None) -> Node:
  return initialize_node("zeros", tuple(shape), name=name, axis_names=axis_names, backend=backend, dtype=dtype)


def ones(shape: Sequence[int], dtype=None, name=None, axis_names=None, backend=None) -> Node:
  return initialize_node("ones", tuple(shape), name=name, axis_names=axis_names, backend=backend, dtype=dtype)


def randn(shape: Sequence[int], dtype=None, seed: Optional[int] = None, name=None, axis_names=None,
          backend=None) -> Node:
  return initialize_node("randn", tuple(shape), name=name, axis_names=axis_names, backend=backend, seed=seed,
                         dtype=dtype)


def random_uniform(shape: Sequence[int], boundaries=(0.0, 1.0), dtype=None, seed: Optional[int] = None, name=None,
                   axis_names=None, backend=None) -> Node:
  return initialize_node("random_uniform", tuple(shape), name=name, axis_names=axis_names, backend=backend,
                         seed=seed, boundaries=boundaries, dtype=dtype)


def norm(node: Node):
  """L2 norm of the node's tensor as a backend scalar (node_linalg.py:214-229)."""
  if not hasattr(node, "backend"):
    raise AttributeError('Node {} of type {} has no `backend`'.format(node, type(node)))
  return node.backend.norm(node.tensor)


def conj(node: Node, name: Optional[str] = None, axis_names: Optional[List[str]] = None) -> Node:
  """Unconnected node holding the complex conjugate (node_linalg.py:232-259)."""
  if not hasattr(node, "backend"):
    raise AttributeError('Node {} of type {} has no `backend`'.format(node, type(node)))
  return Node(node.backend.conj(node.tensor), name=name if name else "conj({})".format(node.name),
              axis_names=list(axis_names) if axis_names else list(node.axis_names), backend=node.backend)


def transpose(node: Node, permutation: Sequence, name: Optional[str] = None,
              axis_names: Optional[List[str]] = None) -> Node:
  """Unconnected node with permuted axes; `permutation` holds axis numbers or names (node_linalg.py:262-294)."""
  if not hasattr(node, "backend"):
    raise AttributeError('Node {} of type {} has no `backend`'.format(node, type(node)))
  perm = [node.get_axis_number(p) for p in permutation]
  out = Node(node.tensor, name=name, axis_names=list(node.axis_names), backend=node.backend).reorder_axes(perm)
  if axis_names:
    out.add_axis_names(list(axis_names))
  return out


def kron(nodes: Sequence[Node]) -> Node:
  """Operator Kronecker product (node_linalg.py:297-331): the outer product of even-order nodes with the
  first halves of all edge lists ("inputs") ahead of the second halves ("outputs"), e.g.
  X_ab, Y_cdef, Z_gh -> R_acdgbefh.  The input nodes are consumed like in any contraction."""
  nodes = list(nodes)
  ins, outs = [], []
  for node in nodes:
    order = len(node.shape)
    if order % 2 != 0:
      raise ValueError(f"All operator tensors must have an even order. Found tensor with order {order}")
    ins += node.edges[:order // 2]
    outs += node.edges[order // 2:]
  result = nodes[0]
  for node in nodes[1:]:
    result = network.outer_product(result, node)
  return result.reorder_edges(ins + outs)
