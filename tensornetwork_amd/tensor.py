"""`Tensor`: a backend array with operator syntax, the object the reference's functional API
(`tn.tensordot`, `tn.svd`, `tn.eye`, ...) passes around (tensor.py:22-202).  For `HipBackend` the wrapped
array is a `DeviceTensor` in HBM; every method is one backend call."""
from typing import Any, Optional, Sequence

import numpy as np


def _resolve_backend(backend):
  from tensornetwork_amd.ncon import _resolve_backend as _rb  # pylint: disable=import-outside-toplevel
  return _rb(backend)


class Tensor:
  """`Tensor(array, backend=None)`; attributes `array`, `backend`, `shape`, `size`, `ndim`, `dtype`."""
  __array_priority__ = 2000   # ndarray (op) Tensor defers to us

  def __init__(self, array: Any, backend=None):
    self.backend = _resolve_backend(backend)
    self.array = self.backend.convert_to_tensor(array)
    self.shape = tuple(self.backend.shape_tuple(self.array))
    self.size = int(np.prod(self.shape, dtype=np.int64))
    self.ndim = len(self.shape)

  @property
  def dtype(self):
    return self.array.dtype

  def _like(self, array) -> "Tensor":
    return Tensor(array, backend=self.backend)

  def _operand(self, other):
    if isinstance(other, Tensor):
      if self.backend.name != other.backend.name:
        raise ValueError(f"Given backends are inconsistent. Found '{self.backend.name}' "
                         f"and '{other.backend.name}'")
      return other.array
    return other

  # ------------------------------------------------------------------ views
  @property
  def T(self) -> "Tensor":      # pylint: disable=invalid-name
    return self.transpose()

  @property
  def H(self) -> "Tensor":      # pylint: disable=invalid-name
    return self._like(self.backend.transpose(self.backend.conj(self.array)))

  def conj(self) -> "Tensor":
    return self._like(self.backend.conj(self.array))

  conjugate = conj

  def hconj(self, perm: Optional[Sequence[int]] = None) -> "Tensor":
    """Complex conjugate with permuted axes (reversed by default)."""
    return self.conj().transpose(perm)

  def transpose(self, perm: Optional[Sequence[int]] = None) -> "Tensor":
    return self._like(self.backend.transpose(self.array, perm))

  def reshape(self, shape: Sequence[int]) -> "Tensor":
    return self._like(self.backend.reshape(self.array, tuple(int(d) for d in shape)))

  def ravel(self) -> "Tensor":
    return self.reshape([self.size])

  def flatten(self) -> "Tensor":
    return self.ravel().copy()

  def squeeze(self) -> "Tensor":
    return self.reshape([d for d in self.shape if d != 1])

  def copy(self) -> "Tensor":
    """A Tensor with its own storage (backends never mutate inputs, so x * 1 is a fresh array)."""
    return self._like(self.backend.multiply(self.array, 1))

  # ------------------------------------------------------------- arithmetic
  def __mul__(self, other):
    return self._like(self.backend.multiply(self.array, self._operand(other)))

  __rmul__ = __mul__

  def __truediv__(self, other):
    return self._like(self.backend.divide(self.array, self._operand(other)))

  def __add__(self, other):
    return self._like(self.backend.addition(self.array, self._operand(other)))

  __radd__ = __add__

  def __sub__(self, other):
    return self._like(self.backend.subtraction(self.array, self._operand(other)))

  def __rsub__(self, other):
    return self._like(self.backend.subtraction(other, self.array))

  def __matmul__(self, other: "Tensor") -> "Tensor":
    if self.backend.name != other.backend.name:
      raise ValueError(f"Backends {self.backend.name} and {other.backend.name} did not agree.")
    return self._like(self.backend.matmul(self.array, other.array))

  def __call__(self, *labels) -> "NconBuilder":
    """`A(1, -1) @ B(1, -2)` builds an ncon call; `tn.finalize` / `NconBuilder.contract` runs it."""
    return NconBuilder([self], [list(labels)])

  def __array__(self, dtype=None, copy=None):  # pylint: disable=unused-argument,redefined-outer-name
    out = np.asarray(self.array)
    return out.astype(dtype) if dtype is not None else out

  def __repr__(self):
    return f"Tensor(shape={self.shape}, dtype={self.dtype}, backend={self.backend.name!r})"


class NconBuilder:
  """Operands and label lists collected by `Tensor.__call__` and `@` (tensor.py:191-202)."""

  def __init__(self, tensors, axes):
    self.tensors = list(tensors)
    self.axes = [list(a) for a in axes]

  def __matmul__(self, other: "NconBuilder") -> "NconBuilder":
    if not isinstance(other, NconBuilder):
      raise TypeError("NconBuilder @ needs another NconBuilder")
    return NconBuilder(self.tensors + other.tensors, self.axes + other.axes)

  def contract(self) -> Tensor:
    from tensornetwork_amd.ncon import ncon  # pylint: disable=import-outside-toplevel
    be = self.tensors[0].backend
    return Tensor(ncon([t.array for t in self.tensors], self.axes, backend=be), backend=be)


def finalize(builder: NconBuilder) -> Tensor:
  """ncon_interface.finalize: run the contraction an `A(...) @ B(...)` expression describes."""
  return builder.contract()
