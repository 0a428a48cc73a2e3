"""Run by tests/test_reference_dropin_cpu.py in a subprocess whose PYTHONPATH holds the reference: google/TensorNetwork's
OWN FiniteMPS / FiniteXXZ / FiniteDMRG classes (matrixproductstates/finite_mps.py, mpo.py, dmrg.py) on backend="hip"
with the library handle bound to the NumPy emulation of the C ABI, beside the same run on the reference's NumPy
backend.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import tensornetwork as tn
import tensornetwork_amd  # noqa: F401  registers "hip"  pylint: disable=unused-import

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import emu_tnh  # noqa: E402  pylint: disable=wrong-import-position
from tensornetwork_amd import _lib  # noqa: E402  pylint: disable=wrong-import-position

_lib._lib, _lib._device = emu_tnh.EmuLib(), 0  # pylint: disable=protected-access


def run(backend, dtype):
  np.random.seed(10)
  n, bond = 10, 16
  mps = tn.FiniteMPS.random([2] * n, [bond] * (n - 1), dtype=dtype, backend=backend)
  mpo = tn.FiniteXXZ(np.ones(n - 1), np.ones(n - 1), np.zeros(n), dtype=dtype, backend=backend)
  energy = tn.FiniteDMRG(mps, mpo).run_one_site(num_sweeps=4, num_krylov_vecs=10, verbose=0)
  sz = np.diag([0.5, -0.5]).astype(dtype)
  local = [float(np.real(np.asarray(x))) for x in mps.measure_local_operator([sz] * n, range(n))]
  corr = [float(np.real(np.asarray(x))) for x in mps.measure_two_body_correlator(sz, sz, 2, [4, 5, 7])]
  mps.position(0)
  norm = float(np.real(np.asarray(mps.backend.norm(mps.tensors[0]))))
  return {"energy": float(np.real(np.asarray(energy))), "sz": local, "szsz": corr, "norm": norm,
          "tensor_type": type(mps.tensors[0]).__name__}


out = {}
for name, dtype in (("float64", np.float64), ("float32", np.float32)):
  out[name] = {"numpy": run("numpy", dtype), "hip": run("hip", dtype)}
print("\nRESULT " + json.dumps(out))
