"""dtype names only (see jax/__init__.py)."""
import numpy as _np

float16, float32, float64 = _np.float16, _np.float32, _np.float64
complex64, complex128 = _np.complex64, _np.complex128
int8, int16, int32, int64 = _np.int8, _np.int16, _np.int32, _np.int64
uint8, uint16, uint32, uint64 = _np.uint8, _np.uint16, _np.uint32, _np.uint64
bool_ = _np.bool_
ndarray = _np.ndarray
