"""NumPy emulation of the C-ABI entry points that HipBackend's CONTRACTION lowering calls.

TEST INFRASTRUCTURE ONLY (like oracle/): the product has no CPU path -- without libtnhip.so and an MI355X every
HipBackend entry point raises.  This module lets the CPU suite (`-m "not gpu"`) drive the backend's HOST logic with
real numbers: the transpose + reshape + GEMM lowering of tensordot (hip_backend._tensordot_impl), the in-place view
lowering (_tensordot_in_place: operand views, the planner's hints, the size gate, the fallbacks), the free-axis
bookkeeping that network.contract_between relies on when it plans layouts, casts, slices.  Each emulated entry point
follows the contract written in include/tnh.h (cited per method); "device" memory is host memory.

Usage (tests/test_host_lowering_cpu.py): `with emulated_backend() as be: ...` -- swaps tensornetwork_amd._lib's
library handle for an EmuLib and restores it afterwards.
"""
import contextlib
import ctypes
import gc

import numpy as np

from tensornetwork_amd import _lib, device_tensor, hip_backend

_NP = {_lib.F32: np.float32, _lib.F64: np.float64, _lib.BF16: np.uint16, _lib.F16: np.float16, _lib.C64: np.complex64,
       _lib.C128: np.complex128, _lib.I32: np.int32, _lib.I64: np.int64}


def _addr(p):
  """int address of a ctypes.c_void_p / int argument."""
  if isinstance(p, int):
    return p
  return p.value or 0


def _ints(arr, n):
  return [int(arr[i]) for i in range(n)]


class EmuLib:
  """The subset of include/tnh.h the contraction lowering uses, on host memory."""

  def __init__(self):
    self._blocks = {}
    self.calls = []            # (entry point, summary) in call order: tests assert on what was launched
    self._last_kernel = b"emu"

  # ---- memory (tnh.h: tnh_malloc / tnh_free / tnh_pool_has / tnh_h2d / tnh_d2h / tnh_d2d / tnh_memset / tnh_sync)
  def tnh_malloc(self, pref, nbytes):
    buf = np.empty(int(nbytes) + 64, dtype=np.uint8)
    base = buf.ctypes.data
    ptr = (base + 63) & ~63
    self._blocks[ptr] = buf
    pref._obj.value = ptr      # pylint: disable=protected-access
    return _lib.OK

  def tnh_free(self, p):
    self._blocks.pop(_addr(p), None)
    return _lib.OK

  def tnh_pool_has(self, nbytes, has_ref):  # pylint: disable=unused-argument
    has_ref._obj.value = 1     # pylint: disable=protected-access
    return _lib.OK

  def tnh_h2d(self, dst, src, nbytes):
    ctypes.memmove(_addr(dst), _addr(src), int(nbytes))
    return _lib.OK

  def tnh_d2h(self, dst, src, nbytes):
    ctypes.memmove(_addr(dst), _addr(src), int(nbytes))
    return _lib.OK

  def tnh_d2d(self, dst, src, nbytes):
    ctypes.memmove(_addr(dst), _addr(src), int(nbytes))
    return _lib.OK

  def tnh_memset(self, dst, byte, nbytes):
    ctypes.memset(_addr(dst), int(byte), int(nbytes))
    return _lib.OK

  def tnh_sync(self):
    return _lib.OK

  def tnh_last_error(self):
    return b""

  def tnh_gemm_last_kernel(self):
    return self._last_kernel

  # ---- helpers
  @staticmethod
  def _flat(ptr, n, dtype):
    """1-D view of n elements of `dtype` at address ptr."""
    dtype = np.dtype(dtype)
    raw = (ctypes.c_uint8 * max(n * dtype.itemsize, 1)).from_address(_addr(ptr))
    return np.frombuffer(raw, dtype=dtype, count=n)

  @staticmethod
  def _to_f(a, code):
    """compute-precision image of stored values (bf16 bit patterns -> float32)."""
    if code == _lib.BF16:
      return device_tensor.bf16_bits_to_f32(a)
    if code == _lib.F16:
      return a.astype(np.float32)
    return a

  @staticmethod
  def _from_f(x, code):
    if code == _lib.BF16:
      return device_tensor.f32_to_bf16_bits(np.asarray(x, dtype=np.float32))
    return np.asarray(x).astype(_NP[code])

  # ---- K1 (tnh.h: dst = numpy.transpose(src, perm), any itemsize)
  def tnh_permute(self, dst, src, rank, shape, perm, itemsize):
    shape, perm = _ints(shape, rank), _ints(perm, rank)
    n = int(np.prod(shape)) if shape else 1
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}.get(int(itemsize))
    if dt is None:
      dt = np.dtype((np.void, int(itemsize)))
    s = self._flat(src, n, dt).reshape(shape)
    out = np.ascontiguousarray(np.transpose(s, perm))
    self._flat(dst, n, dt)[:] = out.reshape(-1)
    self.calls.append(("permute", tuple(shape), tuple(perm)))
    return _lib.OK

  # ---- K1 gather (tnh.h: dst contiguous `shape` gathers src[offset + sum idx_i * stride_i], strides in elements)
  def tnh_strided_copy(self, dst, src, rank, shape, strides, offset, itemsize):
    shape, strides = _ints(shape, rank), _ints(strides, rank)
    n = int(np.prod(shape)) if shape else 1
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[int(itemsize)]
    idx = np.full(shape if shape else (), int(offset), dtype=np.int64)
    for d, (sz, st) in enumerate(zip(shape, strides)):
      ix = np.arange(sz, dtype=np.int64) * st
      idx = idx + ix.reshape([-1 if i == d else 1 for i in range(len(shape))])
    span = int(idx.max()) + 1 if n else 0
    s = self._flat(src, span, dt)
    self._flat(dst, n, dt)[:] = s[idx.reshape(-1)]
    return _lib.OK

  # ---- K6 cast (tnh.h: tnh_cast)
  def tnh_cast(self, dst, dst_code, src, src_code, n):
    n = int(n)
    x = self._to_f(self._flat(src, n, _NP[src_code]), src_code)
    if dst_code in (_lib.I32, _lib.I64) and src_code not in (_lib.I32, _lib.I64):
      x = np.trunc(np.real(x))
    if dst_code not in (_lib.C64, _lib.C128) and np.iscomplexobj(x):
      x = np.real(x)
    self._flat(dst, n, _NP[dst_code])[:] = self._from_f(x, dst_code)
    return _lib.OK

  # ---- K2 (tnh.h: C[b] (M x N, ldc) = op(A[b]) (M x K) * op(B[b]) (K x N); f32 accumulate for f32 / bf16 / f16)
  def _matrix(self, ptr, code, rows, cols, ld, batch_stride, b):
    n = (rows - 1) * ld + cols
    base = _addr(ptr) + b * batch_stride * np.dtype(_NP[code]).itemsize
    flat = self._flat(base, n, _NP[code])
    m = np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(ld * flat.itemsize, flat.itemsize))
    return self._to_f(m, code)

  def tnh_gemm(self, in_code, out_code, trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, batch, sa, sb, sc):
    acc = np.float64 if in_code in (_lib.F64, _lib.C128, _lib.I32, _lib.I64) else np.float32
    if in_code in (_lib.C64,):
      acc = np.complex64
    if in_code in (_lib.C128,):
      acc = np.complex128
    if in_code in (_lib.I32, _lib.I64):
      acc = np.int64
    for bi in range(int(batch)):
      am = self._matrix(a, in_code, k if trans_a else m, m if trans_a else k, lda, sa, bi)
      bm = self._matrix(b, in_code, n if trans_b else k, k if trans_b else n, ldb, sb, bi)
      am = am.T if trans_a else am
      bm = bm.T if trans_b else bm
      prod = np.matmul(am.astype(acc), bm.astype(acc))
      cbase = _addr(c) + bi * sc * np.dtype(_NP[out_code]).itemsize
      cflat = self._flat(cbase, (m - 1) * ldc + n, _NP[out_code])
      cm = np.lib.stride_tricks.as_strided(cflat, shape=(m, n), strides=(ldc * cflat.itemsize, cflat.itemsize))
      cm[:, :] = self._from_f(prod, out_code)
    self._last_kernel = b"emu_gemm"
    self.calls.append(("gemm", int(trans_a), int(trans_b), int(m), int(n), int(k), int(batch)))
    return _lib.OK

  def tnh_gemm_ex(self, in_code, out_code, trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, batch, sa, sb, sc, alpha,
                  beta):
    assert alpha == 1.0 and beta == 0.0
    return self.tnh_gemm(in_code, out_code, trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, batch, sa, sb, sc)

  # ---- K2 view (tnh.h: element (r, k) at (r / r0) sr1 + (r % r0) sr0 + (k / k0) sk1 + (k % k0) sk0; exactly one
  #      of sk0 / sr0 is 1, k0 % 32 == 0, K % 64 == 0, K % k0 == 0, >= 192 tiles of 256 x 256, bf16 / f16)
  def tnh_gemm_view(self, in_code, out_code, m, n, k, a, va, b, vb, c, ldc):
    va, vb = va._obj, vb._obj          # pylint: disable=protected-access
    if in_code not in (_lib.BF16, _lib.F16):
      return _lib.ERR_INVALID
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    if m < 256 or n < 256 or tiles < 192 or k % 64 or k < 128:
      return _lib.ERR_UNSUPPORTED
    mats = []
    for ptr, v, rows in ((a, va, m), (b, vb, n)):
      if v.k0 <= 0 or v.k0 % 32 or k % v.k0 or (v.sk0 == 1) == (v.sr0 == 1) or _addr(ptr) % 16:
        return _lib.ERR_UNSUPPORTED
      r = np.arange(rows, dtype=np.int64)[:, None]
      kk = np.arange(k, dtype=np.int64)[None, :]
      idx = (r // v.r0) * v.sr1 + (r % v.r0) * v.sr0 + (kk // v.k0) * v.sk1 + (kk % v.k0) * v.sk0
      flat = self._flat(ptr, int(idx.max()) + 1, _NP[in_code])
      mats.append(self._to_f(flat[idx], in_code).astype(np.float32))
    prod = mats[0] @ mats[1].T
    cflat = self._flat(c, (m - 1) * ldc + n, _NP[out_code])
    cm = np.lib.stride_tricks.as_strided(cflat, shape=(m, n), strides=(ldc * cflat.itemsize, cflat.itemsize))
    cm[:, :] = self._from_f(prod, out_code)
    a_km, b_kn = va.sk0 != 1, vb.sk0 != 1       # the library's names: tnh_gemm_bf16.hip gemm_bf16_view
    kind = ("tt" if b_kn else "tn") if a_km else ("nn" if b_kn else "nt")
    self._last_kernel = ("emu_view_" + kind).encode()
    self.calls.append(("view_gemm", int(m), int(n), int(k), (va.sk0, va.sr1, va.sk1), (vb.sk0, vb.sr1, vb.sk1)))
    return _lib.OK

  # ---- K6 (tnh.h: dst = a (op) b with broadcasting expressed as element strides) -- used by outer_product
  def tnh_binary(self, op, dst, a, b, rank, shape, a_strides, b_strides, code):
    shape, sa, sb = _ints(shape, rank), _ints(a_strides, rank), _ints(b_strides, rank)
    n = int(np.prod(shape)) if shape else 1

    def operand(ptr, strides):
      idx = np.zeros(shape if shape else (), dtype=np.int64)
      for d, (sz, st) in enumerate(zip(shape, strides)):
        idx = idx + (np.arange(sz, dtype=np.int64) * st).reshape([-1 if i == d else 1 for i in range(len(shape))])
      flat = self._flat(ptr, int(idx.max()) + 1, _NP[code])
      return self._to_f(flat[idx], code)

    x, y = operand(a, sa), operand(b, sb)
    with np.errstate(all="ignore"):
      out = {_lib.OP_ADD: np.add, _lib.OP_SUB: np.subtract, _lib.OP_MUL: np.multiply, _lib.OP_DIV: np.divide}[op](x, y)
    self._flat(dst, n, _NP[code])[:] = self._from_f(out, code).reshape(-1)
    return _lib.OK


  # tnh.h: dst_i = src_i (op) (re + i im), scalar on the left when scalar_left != 0
  def tnh_binary_scalar(self, op, dst, src, re, im, scalar_left, n, code):
    n = int(n)
    x = self._to_f(self._flat(src, n, _NP[code]), code)
    s = complex(re, im) if code in (_lib.C64, _lib.C128) else re
    if code in (_lib.I32, _lib.I64):
      s = int(re)
    fn = {_lib.OP_ADD: np.add, _lib.OP_SUB: np.subtract, _lib.OP_MUL: np.multiply, _lib.OP_DIV: np.divide,
          _lib.OP_POW: np.power}[op]
    with np.errstate(all="ignore"):
      out = fn(s, x) if scalar_left else fn(x, s)
    self._flat(dst, n, _NP[code])[:] = self._from_f(out, code)
    return _lib.OK


class EmulatedHipBackend(hip_backend.HipBackend):
  """HipBackend whose C ABI is the NumPy emulation above (host logic under test, kernels not)."""

  def __init__(self, emu, **kwargs):
    super().__init__(**kwargs)
    self._emu = emu

  @property
  def lib(self):
    return self._emu


@contextlib.contextmanager
def emulated_backend(**kwargs):
  """Context manager: an EmulatedHipBackend with tensornetwork_amd._lib bound to the emulation."""
  saved = (_lib._lib, _lib._device)      # pylint: disable=protected-access
  emu = EmuLib()
  _lib._lib, _lib._device = emu, 0       # pylint: disable=protected-access
  try:
    yield EmulatedHipBackend(emu, **kwargs)
  finally:
    gc.collect()                         # blocks of dead tensors go back through the emulation, not the real library
    _lib._lib, _lib._device = saved      # pylint: disable=protected-access
