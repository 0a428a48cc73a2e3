"""Base class selection for ``HipBackend``.

When google/TensorNetwork is importable, ``HipBackend`` derives from its
``AbstractBackend`` (``tensornetwork/backends/abstract_backend.py:22``) so that
``isinstance`` checks and the factory's object pass-through
(``backend_factory.py:37-38``) work.  When it is not installed (e.g. on a bare
GPU box) an interface mirror with the same method names and the same
"not implemented" error behaviour is used, so ``tensornetwork_amd``'s own
``ncon`` / graph layer keeps working stand-alone.
"""
# pylint: disable=missing-function-docstring

_METHODS = [
    "tensordot", "reshape", "transpose", "slice", "svd", "qr", "rq",
    "shape_concat", "shape_tensor", "shape_tuple", "sparse_shape", "shape_prod",
    "sqrt", "convert_to_tensor", "outer_product", "einsum", "norm", "eye",
    "ones", "zeros", "randn", "random_uniform", "conj", "eigh", "eigs", "eigsh",
    "eigsh_lanczos", "gmres", "addition", "subtraction", "multiply", "divide",
    "index_update", "inv", "broadcast_right_multiplication",
    "broadcast_left_multiplication", "sin", "cos", "exp", "log", "expm", "jit",
    "sum", "matmul", "diagflat", "diagonal", "trace", "abs", "sign",
    "serialize_tensor", "deserialize_tensor", "power", "item", "cholesky", "eps",
]


def _make_stub(name):
  def stub(self, *args, **kwargs):  # pylint: disable=unused-argument
    raise NotImplementedError(
        "Backend '{}' has not implemented {}.".format(self.name, name))
  stub.__name__ = name
  return stub


class _InterfaceMirror:
  """Stand-in for tensornetwork's AbstractBackend (same surface, same errors)."""

  def __init__(self):
    self.name = "abstract backend"

  def pivot(self, tensor, pivot_axis=-1):
    # abstract_backend.py:938-962: reshape to a matrix around pivot_axis.
    ndim = len(self.shape_tuple(tensor))
    if pivot_axis > ndim:
      raise ValueError("Cannot pivot about (zero-indexed) axis {} of a rank-{} tensor."
                       .format(pivot_axis, ndim))
    shape = self.shape_tuple(tensor)
    left, right = 1, 1
    for s in shape[:pivot_axis]:
      left *= s
    for s in shape[pivot_axis:]:
      right *= s
    return self.reshape(tensor, (left, right))


for _name in _METHODS:
  setattr(_InterfaceMirror, _name, _make_stub(_name))

try:  # pragma: no cover - depends on the environment
  from tensornetwork.backends.abstract_backend import AbstractBackend as BackendBase
  HAVE_TENSORNETWORK = True
except Exception:  # pylint: disable=broad-except
  BackendBase = _InterfaceMirror
  HAVE_TENSORNETWORK = False
