"""pytest plugin that runs google/TensorNetwork's OWN test files against this backend.

The reference's root conftest (``/root/reference/conftest.py:16-18``) imports jax and
tensorflow, which are not installed, so the reference tests are collected with
``--noconftest`` and this plugin re-provides what that conftest provides:

  * the ``backend`` argument (``conftest.py:22-25``) -- every test function that takes it is
    parametrised with ``$TNH_REF_BACKENDS`` (default ``"hip"``; ``"numpy"`` validates the
    harness on a machine without a GPU).  Direct parametrisation overrides a fixture of the
    same name, so the module-level fixture of ``tests/ncon_interface_test.py:26-36`` (names AND
    backend objects of numpy/jax/pytorch/tensorflow) is replaced too; for that file the
    backend OBJECT is added as a second parameter, as its own fixture does
    (object pass-through, ``backend_factory.py:37-38``),
  * the autouse default-backend reset (``conftest.py:28-32``).

Tests whose backend list is hard-coded in their own ``@pytest.mark.parametrize`` (numpy / jax /
tensorflow / pytorch) cannot be steered without editing the reference and are deselected, as is
everything that does not take a backend at all: what remains, and is counted, is exactly the set
of reference tests that exercise the backend under test.

Importing ``tensornetwork_amd`` registers ``"hip"`` in ``backend_factory._BACKENDS``
(``tensornetwork_amd.hip_backend.register_with_tensornetwork``), which is all a reference
user has to do.  No reference file is modified.

The reference's TEST HELPERS keep per-backend tables (``tests/testing_utils.py:63-84``
``np_dtype_to_backend`` knows four names; ``linalg/tests/initialization_test.py`` and
``node_linalg_test.py`` keep a module-level ``dtypes`` dict per backend name).  A maintainer who
adds a backend adds its row there; this plugin applies exactly that row
at run time: the hip backend's dtypes ARE NumPy dtypes, so it reuses the ``"numpy"`` rows.
"""
import os

import pytest

import tensornetwork
from tensornetwork.backends import backend_factory
import tensornetwork_amd  # noqa: F401  registers "hip"  pylint: disable=unused-import

_NAMES = [b for b in os.environ.get("TNH_REF_BACKENDS", "hip").split(",") if b]
if os.environ.get("TNH_REF_EMULATED", "0") == "1":
  # CPU run (tests/test_reference_dropin_cpu.py): the backend's HOST code and the C-ABI contract with the reference as
  # the caller -- the library handle is the NumPy emulation of include/tnh.h that the CPU suite uses (tests/emu_tnh.py;
  # test infrastructure, the kernels themselves are covered on the GPU)
  import sys
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
  import emu_tnh  # pylint: disable=import-error,wrong-import-position
  from tensornetwork_amd import _lib as _tnh_lib  # pylint: disable=wrong-import-position
  _tnh_lib._lib, _tnh_lib._device = emu_tnh.EmuLib(), 0  # pylint: disable=protected-access
_KEEP_ALL = os.environ.get("TNH_REF_KEEP_ALL", "0") == "1"


def _teach_test_tables_the_new_backend(module):
  """The per-backend rows a maintainer would add to the reference's test helpers (see module docstring)."""
  try:
    from tensornetwork.tests import testing_utils  # pylint: disable=import-outside-toplevel
  except Exception:  # pylint: disable=broad-except
    testing_utils = None
  if testing_utils is not None and not getattr(testing_utils, "_tnh_patched", False):
    original = testing_utils.np_dtype_to_backend

    def np_dtype_to_backend(backend, dtype):
      name = getattr(backend_factory.get_backend(backend), "name", None)
      if name in _NAMES and name != "numpy":
        return dtype                      # NumPy dtypes, like the "numpy" / "symmetric" row (testing_utils.py:69-70)
      return original(backend, dtype)
    testing_utils.np_dtype_to_backend = np_dtype_to_backend
    testing_utils._tnh_patched = True     # pylint: disable=protected-access
  tables = getattr(module, "dtypes", None)
  if isinstance(tables, dict) and "numpy" in tables:
    for name in _NAMES:
      tables.setdefault(name, tables["numpy"])


def _has_own_backend_parametrize(metafunc):
  for mark in metafunc.definition.iter_markers("parametrize"):
    argnames = mark.args[0]
    if isinstance(argnames, str):
      argnames = [a.strip() for a in argnames.split(",")]
    if "backend" in argnames:
      return True
  return False


def pytest_generate_tests(metafunc):
  _teach_test_tables_the_new_backend(metafunc.module)
  if "backend" not in metafunc.fixturenames or _has_own_backend_parametrize(metafunc):
    return
  values, ids = list(_NAMES), list(_NAMES)
  if metafunc.module.__name__.endswith("ncon_interface_test"):
    for name in _NAMES:
      values.append(backend_factory.get_backend(name))
      ids.append(name + "-object")
  # a parametrised fixture of the same name defined in the test module itself
  # (tests/ncon_interface_test.py:26-36) cannot be overridden by direct parametrisation
  # ("duplicate parametrization"): swap its parameter list instead, before pytest's own
  # fixture parametrisation reads it (plugins run ahead of the core hook implementation).
  defs = getattr(metafunc, "_arg2fixturedefs", {}).get("backend") or []
  if defs and getattr(defs[-1], "params", None) is not None:
    defs[-1].params = values
    defs[-1].ids = ids
    return
  metafunc.parametrize("backend", values, ids=ids)


def _uses_backend_under_test(item):
  callspec = getattr(item, "callspec", None)
  if callspec is None or "backend" not in callspec.params:
    return False
  value = callspec.params["backend"]
  return (value in _NAMES) or (getattr(value, "name", None) in _NAMES)


def pytest_collection_modifyitems(config, items):
  if _KEEP_ALL:
    return
  keep = [it for it in items if _uses_backend_under_test(it)]
  drop = [it for it in items if not _uses_backend_under_test(it)]
  if drop:
    config.hook.pytest_deselected(items=drop)
    items[:] = keep


@pytest.fixture(autouse=True)
def reset_default_backend():
  tensornetwork.set_default_backend("numpy")
  yield
  tensornetwork.set_default_backend("numpy")


def pytest_report_header(config):  # pylint: disable=unused-argument
  return [f"tnh reference drop-in: tensornetwork {tensornetwork.__version__} from "
          f"{os.path.dirname(tensornetwork.__file__)}; backends under test {_NAMES}; "
          f"registered: {sorted(backend_factory._BACKENDS)}"]  # pylint: disable=protected-access
