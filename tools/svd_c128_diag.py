"""Diagnostics of the f64 band path on the real embedding of a complex128 matrix (exact pairs of singular values):
calls the C ABI stages directly, reads the kept brackets / shifts / vectors back and says which check fails."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import tensornetwork_amd as ta  # noqa: E402
from tensornetwork_amd import _lib  # noqa: E402
from tensornetwork_amd.device_tensor import DeviceTensor  # noqa: E402

be = ta.get_hip_backend()
lib = be.lib
rng = np.random.default_rng(13)
m, n, kc = 768, 640, 40
a = rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
# phi(conj(A)): [[re, im], [-im, re]] blocks
e = np.zeros((2 * m, 2 * n))
e[0::2, 0::2], e[0::2, 1::2], e[1::2, 0::2], e[1::2, 1::2] = a.real, -a.imag * -1, -a.imag, a.real
mm, nn, k = 2 * m, 2 * n, 2 * kc
da = be.convert_to_tensor(e)
nbytes = ctypes.c_size_t(0)
_lib.check(lib.tnh_svd_band_work_bytes(_lib.F64, mm, nn, k, ctypes.byref(nbytes)))
work = DeviceTensor.empty((nbytes.value // 8 + 1,), _lib.F64)
s_all = DeviceTensor.empty((nn,), _lib.F64)
st = ctypes.c_int(0)
_lib.check(lib.tnh_svd_band_factor(_lib.F64, mm, nn, ctypes.c_void_p(da.ptr), ctypes.c_void_p(s_all.ptr),
                                   ctypes.c_void_p(work.ptr), k, ctypes.byref(st)), "factor")
print("factor status", st.value)
sr = np.linalg.svd(e, compute_uv=False)
sa = np.asarray(s_all)
print("values err / s0:", np.max(np.abs(sa - sr)) / sr[0], " pair split (ours):", np.max(np.abs(sa[0:2 * kc:2] - sa[1:2 * kc:2])) / sr[0])
offs = (ctypes.c_int64 * 16)()
_lib.check(lib.tnh_svd_band_layout(_lib.F64, mm, nn, k, offs, 16))
base = (work.ptr + 255) & ~255
u = DeviceTensor.empty((mm, k), _lib.F64)
vh = DeviceTensor.empty((k, nn), _lib.F64)
sk = DeviceTensor.empty((k,), _lib.F64)


def peek(off, count, dtype=np.float64):
  host = np.empty(count, dtype=dtype)
  _lib.check(lib.tnh_d2h(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(base + off), host.nbytes))
  return host


for ns in ("1", "0"):
  os.environ["TNH_SVDB_NS"] = ns
  _lib.check(lib.tnh_svd_band_vectors(_lib.F64, mm, nn, ctypes.c_void_p(work.ptr), k, k, ctypes.c_void_p(u.ptr),
                                      ctypes.c_void_p(vh.ptr), ctypes.c_void_p(sk.ptr), ctypes.byref(st)), "vectors")
  print(f"NS={ns}: vectors status", st.value)
  lo, hi = peek(offs[9], nn), peek(offs[10], nn)
  q = nn - 1 - np.arange(k)
  wid = (hi - lo)[q]
  mid = 0.5 * (hi + lo)[q]
  print("  kept bracket widths / s0: max %.2e  median %.2e;  true value inside: %d of %d" % (
      wid.max() / sr[0], np.median(wid) / sr[0], int(np.sum((lo[q] <= sr[:k] * (1 + 1e-15)) & (sr[:k] <= hi[q] * (1 + 1e-15)))), k))
  print("  |mid - true| / s0 max %.2e ; pair midpoint split max %.2e" % (np.max(np.abs(mid - sr[:k])) / sr[0], np.max(np.abs(mid[0::2] - mid[1::2])) / sr[0]))
  sh = peek(offs[12], 2 * k)
  print("  shifts == mid:", np.allclose(sh[:k], mid), " halfwidths max %.2e" % (sh[k:].max() / sr[0]))
  vv = np.asarray(vh)
  g = vv @ vv.T
  print("  final V orth err %.2e   U orth err %.2e" % (np.max(np.abs(g - np.eye(k))), np.max(np.abs(np.asarray(u).T @ np.asarray(u) - np.eye(k)))))
  print("  kept values err %.2e" % (np.max(np.abs(np.asarray(sk) - sr[:k])) / sr[0]))
  if ns == "0":
    x = peek(offs[11], k * nn).reshape(k, nn)           # band vectors after the cluster MGS (no NS)
    gx = x @ x.T
    off = np.abs(gx - np.eye(k))
    i, j = np.unravel_index(np.argmax(off), off.shape)
    print("  band X orth err %.2e at (%d, %d); pair blocks max %.2e; others max %.2e" % (
        off.max(), i, j, max(off[2 * t, 2 * t + 1] for t in range(kc)),
        np.max(off - np.kron(np.eye(kc), np.ones((2, 2))) * off)))
