import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor
be = ta.get_hip_backend()
rng = np.random.default_rng(13)
m, n, kc = 768, 640, 40
a = rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
mine = np.zeros((2 * m, 2 * n))
mine[0::2, 0::2], mine[0::2, 1::2], mine[1::2, 0::2], mine[1::2, 1::2] = a.real, a.imag, -a.imag, a.real
theirs = np.zeros((2 * m, 2 * n))
theirs[0::2, 0::2], theirs[0::2, 1::2], theirs[1::2, 0::2], theirs[1::2, 1::2] = a.real, -a.imag, a.imag, a.real
emb = DeviceTensor.empty((2 * m, 2 * n), _lib.F64)
ad = be.convert_to_tensor(a)
_lib.check(be.lib.tnh_complex_expand(ctypes.c_void_p(emb.ptr), ctypes.c_void_p(ad.ptr), m, n, n, 1, 1, _lib.C128))
print("expand == theirs:", np.array_equal(np.asarray(emb), theirs), " == mine:", np.array_equal(np.asarray(emb), mine))
for name, mat in (("mine", be.convert_to_tensor(mine)), ("theirs", be.convert_to_tensor(theirs)), ("expand", emb)):
  for rep in range(3):
    out = be._svd_band(mat, 2 * m, 2 * n, 2 * kc, None, False)
    print(name, rep, "ok" if out is not None else "None", be.last_svd_path, be.last_svd_band_status)
for rep in range(2):
  out = be.svd(ad, 1, max_singular_values=kc)
  print("svd c128", rep, be.last_svd_path, be.last_svd_band_status)
g = be.convert_to_tensor(rng.standard_normal((1536, 1280)))
for rep in range(3):
  out = be._svd_band(g, 1536, 1280, 80, None, False)
  print("gauss f64", rep, "ok" if out is not None else "None", be.last_svd_band_status)
