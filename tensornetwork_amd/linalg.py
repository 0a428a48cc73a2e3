"""Functional API on `Tensor` objects: the reference's `linalg/operations.py`, `linalg/linalg.py`,
`linalg/initialization.py` and `linalg/krylov.py` restated as thin one-call wrappers (each function names
the backend method it forwards to).  With `HipBackend` all of it runs on device tensors."""
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np

from tensornetwork_amd.tensor import Tensor, _resolve_backend


def _same_backend(tensors: Sequence[Tensor], fname: str):
  names = [t.backend.name for t in tensors]
  if any(n != names[0] for n in names[1:]):
    raise ValueError("All Tensors fed to " + fname + "must have the same backend.Backends were: \n" +
                     str([n + "\n" for n in names]))
  return tensors[0].backend


def _unary(method: str, doc: str) -> Callable[[Tensor], Tensor]:
  def fun(tensor: Tensor) -> Tensor:
    return Tensor(getattr(tensor.backend, method)(tensor.array), backend=tensor.backend)
  fun.__name__ = method
  fun.__doc__ = doc
  return fun


# ---------------------------------------------------------------- operations.py
def tensordot(a: Tensor, b: Tensor, axes) -> Tensor:
  """operations.py:40-62 -> backend.tensordot (the hot path)."""
  if a.backend.name != b.backend.name:
    raise ValueError("Tried to Tensordot Tensors with differing backends \n" + a.backend.name + "and " +
                     b.backend.name + ".")
  return Tensor(a.backend.tensordot(a.array, b.array, axes), backend=a.backend)


def reshape(tensor: Tensor, new_shape: Sequence[int]) -> Tensor:
  return tensor.reshape(new_shape)


def transpose(tensor: Tensor, perm: Optional[Sequence[int]] = None) -> Tensor:
  return tensor.transpose(perm)


def hconj(tensor: Tensor, perm: Optional[Sequence[int]] = None) -> Tensor:
  return tensor.hconj(perm)


def conj(tensor: Tensor) -> Tensor:
  return tensor.conj()


def take_slice(tensor: Tensor, start_indices: Tuple[int, ...], slice_sizes: Tuple[int, ...]) -> Tensor:
  """operations.py:88-101 -> backend.slice."""
  return Tensor(tensor.backend.slice(tensor.array, start_indices, slice_sizes), backend=tensor.backend)


def shape(tensor: Tensor) -> Tuple[int, ...]:
  return tensor.shape


def outer(tensor1: Tensor, tensor2: Tensor) -> Tensor:
  be = _same_backend([tensor1, tensor2], "outer")
  return Tensor(be.outer_product(tensor1.array, tensor2.array), backend=be)


def einsum(expression: str, *tensors: Tensor, optimize: bool = True) -> Tensor:  # pylint: disable=unused-argument
  """operations.py:131-139; pairwise contractions through `tensornetwork_amd.ncon.einsum`."""
  from tensornetwork_amd.ncon import einsum as _einsum  # pylint: disable=import-outside-toplevel
  be = _same_backend(list(tensors), "einsum")
  return Tensor(_einsum(expression, *[t.array for t in tensors], backend=be), backend=be)


sqrt = _unary("sqrt", "Elementwise square root (operations.py:115-118).")
sin = _unary("sin", "Elementwise sine (operations.py:165-174).")
cos = _unary("cos", "Elementwise cosine (operations.py:177-186).")
exp = _unary("exp", "Elementwise exponential (operations.py:189-198).")
log = _unary("log", "Elementwise natural logarithm (operations.py:201-210).")
sign = _unary("sign", "Elementwise sign (operations.py:270-276).")
abs = _unary("abs", "Elementwise absolute value (operations.py:279-284).")  # pylint: disable=redefined-builtin


def diagonal(tensor: Tensor, offset: int = 0, axis1: int = -2, axis2: int = -1) -> Tensor:
  return Tensor(tensor.backend.diagonal(tensor.array, offset=offset, axis1=axis1, axis2=axis2),
                backend=tensor.backend)


def diagflat(tensor: Tensor, k: int = 0) -> Tensor:
  return Tensor(tensor.backend.diagflat(tensor.array, k=k), backend=tensor.backend)


def trace(tensor: Tensor, offset: int = 0, axis1: int = -2, axis2: int = -1) -> Tensor:
  return Tensor(tensor.backend.trace(tensor.array, offset=offset, axis1=axis1, axis2=axis2),
                backend=tensor.backend)


def pivot(tensor: Tensor, pivot_axis: int = -1) -> Tensor:
  return Tensor(tensor.backend.pivot(tensor.array, pivot_axis=pivot_axis), backend=tensor.backend)


def kron(tensorA: Tensor, tensorB: Tensor) -> Tensor:  # pylint: disable=invalid-name
  """Tensor Kronecker product of two even-order tensors (operations.py:300-342): reshaped about the
  middle it equals np.kron of the two matrices; index order (inA..., inB..., outA..., outB...)."""
  be = _same_backend([tensorA, tensorB], "kron")
  na, nb = tensorA.ndim, tensorB.ndim
  for name, n in (("tensorA", na), ("tensorB", nb)):
    if n % 2 != 0:
      raise ValueError(f"kron only supports tensors with even number of legs.found {name}.ndim = {n}")
  perm = list(range(na // 2)) + list(range(na, na + nb // 2)) + list(range(na // 2, na)) + \
      list(range(na + nb // 2, na + nb))
  return Tensor(be.transpose(be.outer_product(tensorA.array, tensorB.array), perm), backend=be)


# -------------------------------------------------------------------- linalg.py
def svd(tensor: Tensor, pivot_axis: int = -1, max_singular_values: Optional[int] = None,
        max_truncation_error: Optional[float] = None, relative: bool = False):
  """linalg.py:19-81 -> backend.svd; returns Tensors (u, s, vh, s_rest)."""
  be = tensor.backend
  out = be.svd(tensor.array, pivot_axis, max_singular_values=max_singular_values,
               max_truncation_error=max_truncation_error, relative=relative)
  return tuple(Tensor(t, backend=be) for t in out)


def qr(tensor: Tensor, pivot_axis: int = -1, non_negative_diagonal: bool = False):
  be = tensor.backend
  return tuple(Tensor(t, backend=be) for t in be.qr(tensor.array, pivot_axis, non_negative_diagonal))


def rq(tensor: Tensor, pivot_axis: int = -1, non_negative_diagonal: bool = False):
  be = tensor.backend
  return tuple(Tensor(t, backend=be) for t in be.rq(tensor.array, pivot_axis, non_negative_diagonal))


def eigh(matrix: Tensor):
  be = matrix.backend
  return tuple(Tensor(t, backend=be) for t in be.eigh(matrix.array))


def norm(tensor: Tensor):
  """L2 norm as a backend scalar (linalg.py:193-198)."""
  return tensor.backend.norm(tensor.array)


def inv(matrix: Tensor) -> Tensor:
  return Tensor(matrix.backend.inv(matrix.array), backend=matrix.backend)


def expm(matrix: Tensor) -> Tensor:
  return Tensor(matrix.backend.expm(matrix.array), backend=matrix.backend)


# ------------------------------------------------------------ initialization.py
def initialize_tensor(fname: str, *fargs: Any, backend=None, **fkwargs: Any) -> Tensor:
  be = _resolve_backend(backend)
  return Tensor(getattr(be, fname)(*fargs, **fkwargs), backend=be)


def eye(N: int, dtype=None, M: Optional[int] = None, backend=None) -> Tensor:  # pylint: disable=invalid-name
  return initialize_tensor("eye", N, backend=backend, dtype=dtype, M=M)


def zeros(shape: Sequence[int], dtype=None, backend=None) -> Tensor:  # pylint: disable=redefined-outer-name
  return initialize_tensor("zeros", tuple(shape), backend=backend, dtype=dtype)


def ones(shape: Sequence[int], dtype=None, backend=None) -> Tensor:  # pylint: disable=redefined-outer-name
  return initialize_tensor("ones", tuple(shape), backend=backend, dtype=dtype)


def _like(fname, tensor, dtype, backend):
  if isinstance(tensor, Tensor):
    be = tensor.backend if backend is None else _resolve_backend(backend)
    shp, dt = tensor.shape, (tensor.dtype if dtype is None else dtype)
  else:
    be = _resolve_backend(backend)
    arr = np.asarray(tensor)
    shp, dt = arr.shape, (arr.dtype if dtype is None else dtype)
  return initialize_tensor(fname, tuple(shp), backend=be, dtype=dt)


def ones_like(tensor, dtype=None, backend=None) -> Tensor:
  return _like("ones", tensor, dtype, backend)


def zeros_like(tensor, dtype=None, backend=None) -> Tensor:
  return _like("zeros", tensor, dtype, backend)


def randn(shape: Sequence[int], dtype=None, seed: Optional[int] = None, backend=None) -> Tensor:  # pylint: disable=redefined-outer-name
  return initialize_tensor("randn", tuple(shape), backend=backend, seed=seed, dtype=dtype)


def random_uniform(shape: Sequence[int], boundaries: Tuple[float, float] = (0.0, 1.0), dtype=None,  # pylint: disable=redefined-outer-name
                   seed: Optional[int] = None, backend=None) -> Tensor:
  return initialize_tensor("random_uniform", tuple(shape), backend=backend, seed=seed, boundaries=boundaries,
                           dtype=dtype)


# -------------------------------------------------------------------- krylov.py
def _krylov_checks(backend, x0, args):
  """krylov.py:55-110: backend from x0 unless given, args unwrapped to backend arrays."""
  if backend is None and x0 is None:
    raise ValueError("One of backend or x0 must be specified.")
  be = x0.backend if backend is None else _resolve_backend(backend)
  if x0 is not None and x0.backend.name != be.name:
    raise ValueError("If both x0 and backend are specified the backends must agree.")
  args = [] if args is None else list(args)
  for a in args:
    if isinstance(a, Tensor) and a.backend.name != be.name:
      raise ValueError("Backend mismatch in args.")
  return be, args


def _array_operator(A: Callable, be):  # pylint: disable=invalid-name
  """`A` maps Tensors to a Tensor; the backend solvers want a function of backend arrays."""
  def op(x, *arrays):
    return A(Tensor(x, backend=be), *[Tensor(a, backend=be) for a in arrays]).array
  return op


def _arrays(args):
  return [a.array if isinstance(a, Tensor) else a for a in args]


def eigsh_lanczos(A: Callable, backend=None, args: Optional[List[Tensor]] = None, x0: Optional[Tensor] = None,  # pylint: disable=invalid-name
                  shape: Optional[Tuple[int, ...]] = None, dtype=None, num_krylov_vecs: int = 20, numeig: int = 1,
                  tol: float = 1e-8, delta: float = 1e-8, ndiag: int = 20, reorthogonalize: bool = False):
  """Lanczos for a Hermitian operator on Tensors (krylov.py:113-173) -> backend.eigsh_lanczos."""
  be, args = _krylov_checks(backend, x0, args)
  vals, vecs = be.eigsh_lanczos(_array_operator(A, be), _arrays(args), None if x0 is None else x0.array, shape,
                                dtype, num_krylov_vecs, numeig, tol, delta, ndiag, reorthogonalize)
  return vals, [Tensor(v, backend=be) for v in vecs]


def eigs(A: Callable, backend=None, args: Optional[List[Tensor]] = None, x0: Optional[Tensor] = None,  # pylint: disable=invalid-name
         shape: Optional[Tuple[int, ...]] = None, dtype=None, num_krylov_vecs: int = 20, numeig: int = 1,
         tol: float = 1e-8, which: str = 'LR', maxiter: int = 20):
  """Arnoldi for a general operator on Tensors (krylov.py:176-261) -> backend.eigs."""
  be, args = _krylov_checks(backend, x0, args)
  vals, vecs = be.eigs(_array_operator(A, be), _arrays(args), None if x0 is None else x0.array, shape, dtype,
                       num_krylov_vecs, numeig, tol, which, maxiter)
  return vals, [Tensor(v, backend=be) for v in vecs]


def gmres(A_mv: Callable, b: Tensor, A_args: Optional[List] = None, x0: Optional[Tensor] = None, tol: float = 1e-5,  # pylint: disable=invalid-name
          atol: Optional[float] = None, num_krylov_vectors: Optional[int] = None, maxiter: int = 1,
          M: Optional[Callable] = None):  # pylint: disable=invalid-name
  """GMRES on Tensors (krylov.py:264-361) -> backend.gmres; returns (x, info)."""
  be, A_args = _krylov_checks(None, b, A_args)  # pylint: disable=invalid-name
  if x0 is not None and x0.backend.name != be.name:
    raise ValueError("x0 and b must have the same backend.")
  kwargs = {} if num_krylov_vectors is None else {"num_krylov_vectors": num_krylov_vectors}
  x, info = be.gmres(_array_operator(A_mv, be), b.array, A_args=_arrays(A_args), x0=None if x0 is None else x0.array,
                     tol=tol, atol=atol, maxiter=maxiter, M=M, **kwargs)
  return Tensor(x, backend=be), info
