"""Runs inside a subprocess with the ASAN runtime preloaded (tests/test_cabi.py): the shim's host code under
AddressSanitizer -- planners, layout arithmetic, argument checks and error paths that need no GPU."""
import ctypes, os, sys
sys.path.insert(0, os.environ["TNH_REPO"])
from tensornetwork_amd import _lib
_lib.LIB_PATH = os.path.join(os.environ["TNH_REPO"], "tensornetwork_amd", "libtnhip_asan.so")
lib = _lib.load_library()
# host-only planners and argument checks: no GPU, no tnh_init
import numpy as np
nb = 8
pairs = (ctypes.c_int32 * ((nb - 1) * (nb // 2) * 2))()
rounds = ctypes.c_int(0)
assert lib.tnh_svd_block_schedule(nb, 1, pairs, ctypes.byref(rounds)) == 0 and rounds.value == nb - 1
sz = ctypes.c_size_t(0)
assert lib.tnh_svd_band_work_bytes(_lib.F32, 4096, 4096, 256, ctypes.byref(sz)) == 0 and sz.value > 0
assert lib.tnh_svd_band_work_bytes(_lib.F32, 100, 4096, 256, ctypes.byref(sz)) != 0      # m < n: refused, error text set
assert b"unsupported" in lib.tnh_last_error()
off = (ctypes.c_int64 * 16)()
assert lib.tnh_svd_band_layout(_lib.F64, 2048, 1024, 64, off, 16) == 0
assert lib.tnh_svd_band_supported(_lib.F32, 1024, 1024, 64) == 1 and lib.tnh_svd_band_supported(_lib.F32, 1024, 1000, 64) == 0
assert lib.tnh_svd_band_last_stage1() == 0
assert lib.tnh_gemm(_lib.F32, _lib.F32, 0, 0, 4, 4, 4, None, 4, None, 4, None, 4, 1, 0, 0, 0) != 0     # not initialised: an error, not a crash
assert lib.tnh_masked_scatter(None, None, None, None, 0, 10, 4, None) != 0
assert lib.tnh_gemm_set_variant(b"no_such_variant") != 0
assert lib.tnh_gemm_set_variant(b"auto") == 0
print("asan drive ok")
