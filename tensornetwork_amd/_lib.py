"""ctypes binding of ``libtnhip.so`` (the C ABI declared in ``include/tnh.h``).

The product path has no CPU fallback: if the shared library is missing, or no
gfx950 device can be initialised, every entry point raises ``HipRuntimeError``.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int32,
                    c_int64, c_size_t, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TNHIP_LIBRARY", os.path.join(_HERE, "libtnhip.so"))


class HipRuntimeError(RuntimeError):
  """libtnhip is missing, failed to initialise, or a HIP call failed."""


# tnh_status (include/tnh.h)
OK, ERR_HIP, ERR_INVALID, ERR_UNSUPPORTED, ERR_NOMEM, ERR_NOT_INIT, \
    ERR_NO_CONVERGE, ERR_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7

# tnh_dtype
F32, F64, BF16, F16, C64, C128, I32, I64 = range(8)

# tnh_unary_op / tnh_binary_op
(OP_SQRT, OP_CONJ, OP_ABS, OP_SIGN, OP_EXP, OP_LOG, OP_SIN, OP_COS, OP_NEG,
 OP_COPY, OP_REAL, OP_IMAG) = range(12)
OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_POW = range(5)

class OperandView(ctypes.Structure):
  """tnh_operand_view (include/tnh.h): two-level row / contraction strides of one GEMM operand."""
  _fields_ = [("r0", c_int64), ("sr0", c_int64), ("sr1", c_int64),
              ("k0", c_int64), ("sk0", c_int64), ("sk1", c_int64)]


GATHER_MAX_DIGITS, GATHER_MAX_TILE_DIGITS = 8, 6

class GatherDesc(ctypes.Structure):
  """tnh_gather_desc (include/tnh.h): tile plan of the long operand of tnh_gemm_gather."""
  _fields_ = [("nd", c_int32), ("ext", c_int32 * GATHER_MAX_DIGITS), ("stride", c_int32 * GATHER_MAX_DIGITS),
              ("mult", c_int32 * GATHER_MAX_DIGITS), ("k_mask", c_int32), ("nt", c_int32),
              ("text", c_int32 * GATHER_MAX_TILE_DIGITS), ("tstride", c_int64 * GATHER_MAX_TILE_DIGITS),
              ("kl_ext", c_int32), ("kl_stride", c_int64)]


_I64P = POINTER(c_int64)
_I32P = POINTER(c_int32)

# name -> (restype, argtypes); every symbol include/tnh.h declares.
SIGNATURES = {
    "tnh_init": (c_int, [c_int]),
    "tnh_shutdown": (c_int, []),
    "tnh_device_count": (c_int, [POINTER(c_int)]),
    "tnh_device_info": (c_int, [c_char_p, c_int, POINTER(c_int), _I64P]),
    "tnh_device_pci_bus_id": (c_int, [c_char_p, c_int]),
    "tnh_last_error": (c_char_p, []),
    "tnh_version": (c_char_p, []),
    "tnh_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "tnh_free": (c_int, [c_void_p]),
    "tnh_trim": (c_int, []),
    "tnh_mem_stats": (c_int, [_I64P, _I64P, _I64P]),
    "tnh_pool_has": (c_int, [ctypes.c_size_t, ctypes.POINTER(c_int)]),
    "tnh_h2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "tnh_d2h": (c_int, [c_void_p, c_void_p, c_size_t]),
    "tnh_d2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "tnh_memset": (c_int, [c_void_p, c_int, c_size_t]),
    "tnh_sync": (c_int, []),
    "tnh_stream": (c_int, [POINTER(c_void_p)]),
    "tnh_event_create": (c_int, [POINTER(c_void_p)]),
    "tnh_event_record": (c_int, [c_void_p]),
    "tnh_event_sync": (c_int, [c_void_p]),
    "tnh_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "tnh_event_destroy": (c_int, [c_void_p]),
    "tnh_graph_begin": (c_int, []),
    "tnh_graph_end": (c_int, [POINTER(c_void_p)]),
    "tnh_graph_launch": (c_int, [c_void_p]),
    "tnh_graph_destroy": (c_int, [c_void_p]),
    "tnh_permute": (c_int, [c_void_p, c_void_p, c_int, _I64P, _I32P, c_int]),
    "tnh_strided_copy": (c_int, [c_void_p, c_void_p, c_int, _I64P, _I64P,
                                 c_int64, c_int]),
    "tnh_strided_scatter": (c_int, [c_void_p, c_void_p, c_int, _I64P, _I64P,
                                    c_int64, c_int]),
    "tnh_gemm": (c_int, [c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64,
                         c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                         c_int64, c_int64, c_int64, c_int64, c_int64]),
    "tnh_gemm_ex": (c_int, [c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64,
                            c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                            c_int64, c_int64, c_int64, c_int64, c_int64, c_double, c_double]),
    "tnh_gemm_gather": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                POINTER(GatherDesc), c_void_p, c_int64, c_int]),
    "tnh_gemm_gather_plan": (c_int, [POINTER(GatherDesc), c_int64, c_int64, c_int64, _I32P, _I32P, _I32P, c_int64,
                                     _I64P, c_int64]),
    "tnh_gemm_view": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, POINTER(OperandView),
                              c_void_p, POINTER(OperandView), c_void_p, c_int64]),
    "tnh_complex_expand": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int]),
    "tnh_gemm_last_kernel": (c_char_p, []),
    "tnh_gemm_set_variant": (c_int, [c_char_p]),
    "tnh_trace_last2": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                c_int64, c_int]),
    "tnh_sum_mid": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64,
                            c_int]),
    "tnh_norm": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "tnh_unary": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int]),
    "tnh_binary": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, _I64P,
                           _I64P, _I64P, c_int]),
    "tnh_binary_scalar": (c_int, [c_int, c_void_p, c_void_p, c_double,
                                  c_double, c_int, c_int64, c_int]),
    "tnh_fill": (c_int, [c_void_p, c_double, c_double, c_int64, c_int]),
    "tnh_compare": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_int64, c_int]),
    "tnh_masked_fill": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_double, c_int64, c_int]),
    "tnh_masked_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, POINTER(c_int64)]),
    "tnh_random": (c_int, [c_void_p, c_int64, c_int, ctypes.c_uint64, c_int, c_double, c_double]),
    "tnh_eye": (c_int, [c_void_p, c_int64, c_int64, c_int]),
    "tnh_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64]),
    "tnh_wrap_int": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int]),
    "tnh_svd_work_bytes": (c_int, [c_int, c_int64, c_int64,
                                   POINTER(c_size_t)]),
    "tnh_svd_factor": (c_int, [c_int, c_int64, c_int64, c_void_p, c_void_p,
                               c_void_p, POINTER(c_int)]),
    "tnh_svd_vectors": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64,
                                c_void_p, c_void_p]),
    "tnh_svd": (c_int, [c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                        c_void_p, c_int64, c_void_p, POINTER(c_int)]),
    "tnh_svd_factor_topk": (c_int, [c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, POINTER(c_int),
                                    POINTER(c_int)]),
    "tnh_svd_block_schedule": (c_int, [c_int, c_int, c_void_p, POINTER(c_int)]),
    "tnh_svd_vectors_topk": (c_int, [c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                     c_void_p]),
    "tnh_svd_band_supported": (c_int, [c_int, c_int64, c_int64, c_int64]),
    "tnh_svd_band_work_bytes": (c_int, [c_int, c_int64, c_int64, c_int64, POINTER(c_size_t)]),
    "tnh_svd_band_layout": (c_int, [c_int, c_int64, c_int64, c_int64, _I64P, c_int]),
    "tnh_svd_band_factor": (c_int, [c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, POINTER(c_int)]),
    "tnh_svd_band_vectors": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                     POINTER(c_int)]),
    "tnh_svd_band_last_stage1": (c_int, []),
    "tnh_qr_work_bytes": (c_int, [c_int, c_int64, c_int64, POINTER(c_size_t)]),
    "tnh_qr": (c_int, [c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tnh_comm_available": (c_int, []),
    "tnh_comm_abort": (c_int, []),
    "tnh_comm_unique_id": (c_int, [c_void_p]),
    "tnh_comm_init": (c_int, [c_void_p, c_int, c_int]),
    "tnh_comm_info": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "tnh_comm_destroy": (c_int, []),
    "tnh_allreduce": (c_int, [c_void_p, c_int64, c_int, c_int]),
    "tnh_allreduce_sum": (c_int, [c_void_p, c_int64, c_int]),
    "tnh_allgather": (c_int, [c_void_p, c_void_p, c_int64]),
    "tnh_broadcast": (c_int, [c_void_p, c_int64, c_int]),
}

_lib = None
_device = None


def load_library():
  """dlopen libtnhip.so and attach the C signatures (no GPU needed)."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise HipRuntimeError(
        f"{LIB_PATH} not found: build it with `python -c 'import "
        f"__graft_entry__ as g; g.build()'` or `make -C tensornetwork_amd/csrc`."
        " The hip backend has no CPU fallback.")
  try:
    lib = ctypes.CDLL(LIB_PATH)
  except OSError as exc:
    raise HipRuntimeError(f"cannot load {LIB_PATH}: {exc}") from exc
  for name, (restype, argtypes) in SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
  _lib = lib
  return lib


def last_error():
  msg = load_library().tnh_last_error()
  return msg.decode("utf-8", "replace") if msg else ""


def check(status, what=""):
  """Map a tnh_status to the Python exception the reference's callers expect."""
  if status == OK:
    return
  msg = f"{what}: {last_error()}" if what else last_error()
  if status == ERR_INVALID:
    raise ValueError(msg)
  if status == ERR_UNSUPPORTED:
    raise NotImplementedError(msg)
  if status == ERR_NOMEM:
    raise MemoryError(msg)
  if status == ERR_TIMEOUT:
    raise TimeoutError(msg)
  raise HipRuntimeError(msg)


def device_count():
  n = c_int(0)
  check(load_library().tnh_device_count(ctypes.byref(n)))
  return n.value


def init(device=None):
  """Bind this process to one MI355X (default: $LOCAL_RANK, else 0)."""
  global _device
  lib = load_library()
  if _device is not None:
    return lib
  if device is None:
    device = int(os.environ.get("TNHIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    n = device_count()
    if n == 0:
      raise HipRuntimeError(
          "no HIP device visible: the hip backend runs on MI355X (gfx950) only "
          "and has no CPU fallback")
    device %= n
  status = lib.tnh_init(int(device))
  if status != OK:
    raise HipRuntimeError(f"tnh_init({device}) failed: {last_error()}")
  _device = int(device)
  return lib


def lib():
  """The initialised library (initialises on first use)."""
  if _device is None:
    return init()
  return _lib


def current_device():
  return _device


def sync():
  check(lib().tnh_sync(), "tnh_sync")


def i64_array(values):
  values = [int(v) for v in values]
  return (c_int64 * max(len(values), 1))(*values)


def i32_array(values):
  values = [int(v) for v in values]
  return (c_int32 * max(len(values), 1))(*values)


class Event:
  """HIP event on the library's stream (kernel timing in bench.py)."""

  def __init__(self):
    self._h = c_void_p()
    check(lib().tnh_event_create(ctypes.byref(self._h)))

  def record(self):
    check(_lib.tnh_event_record(self._h))
    return self

  def synchronize(self):
    check(_lib.tnh_event_sync(self._h))

  def elapsed_ms(self, later):
    ms = c_float(0.0)
    check(_lib.tnh_event_elapsed_ms(self._h, later._h, ctypes.byref(ms)))
    return ms.value

  def __del__(self):
    try:
      if self._h and _lib is not None:
        _lib.tnh_event_destroy(self._h)
    except Exception:  # pylint: disable=broad-except
      pass
