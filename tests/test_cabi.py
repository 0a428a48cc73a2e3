"""The C-ABI shared library: it loads, exports every symbol include/tnh.h
declares, the ctypes table covers them all, and the product path fails loudly
(no CPU fallback) when no gfx950 device is present."""
import os
import re

import pytest

from tensornetwork_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
  text = open(os.path.join(REPO, "include", "tnh.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(tnh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
  lib = _lib.load_library()
  names = header_symbols()
  assert len(names) >= 40
  for name in names:
    assert hasattr(lib, name), f"{name} declared in include/tnh.h but not exported"


def test_ctypes_table_matches_header():
  assert sorted(_lib.SIGNATURES) == header_symbols()


def test_version_and_error_strings():
  lib = _lib.load_library()
  assert b"gfx950" in lib.tnh_version()
  assert isinstance(_lib.last_error(), str)


def test_status_codes_map_to_reference_exception_types():
  # ValueError / NotImplementedError / MemoryError, never abort
  with pytest.raises(ValueError):
    _lib.check(_lib.ERR_INVALID, "x")
  with pytest.raises(NotImplementedError):
    _lib.check(_lib.ERR_UNSUPPORTED, "x")
  with pytest.raises(MemoryError):
    _lib.check(_lib.ERR_NOMEM, "x")
  with pytest.raises(_lib.HipRuntimeError):
    _lib.check(_lib.ERR_HIP, "x")


def test_calls_before_init_return_status_not_crash():
  lib = _lib.load_library()
  if _lib.current_device() is not None:
    pytest.skip("library already initialised in this process")
  assert lib.tnh_sync() == _lib.ERR_NOT_INIT
  assert "tnh_init" in _lib.last_error()


def test_no_gpu_means_loud_failure_not_fallback():
  if _lib.device_count() > 0:
    pytest.skip("a GPU is visible")
  import numpy as np
  import tensornetwork_amd as ta
  with pytest.raises(ta.HipRuntimeError):
    ta.HipBackend().convert_to_tensor(np.ones(3))
  with pytest.raises(ta.HipRuntimeError):
    ta.ncon([np.ones((2, 2)), np.ones((2, 2))], [(-1, 1), (1, -2)])


def test_product_package_never_imports_the_oracle():
  pkg = os.path.join(REPO, "tensornetwork_amd")
  for root, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".h")):
        text = open(os.path.join(root, f)).read()
        assert "numpy_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_host_entry_points_under_asan():
  """SURVEY section 5 (VERDICT r5 missing 7): the shim's HOST code under AddressSanitizer.  `make -C tools/asan`
  builds libtnhip_asan.so (device code as usual: GPU ASAN is not available on this pool);
  a subprocess with the ASAN runtime preloaded drives the entry points that need no GPU -- the block-Jacobi schedule,
  the band SVD's workspace / layout arithmetic, argument checks and error paths.  Skipped when the ASAN library has not
  been built (`__graft_entry__.build()` builds it best-effort)."""
  import glob
  import os
  import subprocess
  import sys
  import pytest
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  lib = os.path.join(root, "tensornetwork_amd", "libtnhip_asan.so")
  rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
  if not os.path.exists(lib) or not rt:
    pytest.skip("libtnhip_asan.so / the ASAN runtime is not there (make -C tools/asan)")
  env = dict(os.environ, TNH_REPO=root, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
  out = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "asan_drive.py")], env=env,
                       capture_output=True, text=True, timeout=300)
  assert out.returncode == 0 and "asan drive ok" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
