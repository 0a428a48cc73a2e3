"""The reference's OWN test files, unmodified, on ``backend="hip"``.

google/TensorNetwork is not installed on the GPU box; when a copy of its package is reachable
(``$TN_REFERENCE_DIR`` or ``<repo>/_reference_scratch``, as rounds 2-5 shipped it with
``tools/reference_dropin/gpurun_with_reference.sh``; round 6 ships nothing) every file below is
run in its own pytest process through ``tools/reference_dropin/tnh_ref_plugin.py``.  A reference
test may fail only for a reason that also makes it fail on the reference's NumPy backend in this
image (``EXPECTED`` lists them with the reason); everything else must pass on the GPU."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_dir():
  # (never /root/reference: the GPU suite does not read the build container's copy -- its CPU counterpart,
  #  tests/test_reference_dropin_cpu.py, is where the reference is the caller in round 6)
  for cand in (os.environ.get("TN_REFERENCE_DIR"), os.path.join(REPO, "_reference_scratch")):
    if cand and os.path.isdir(os.path.join(cand, "tensornetwork")):
      return cand
  return None


# file -> regexes of test ids that fail for reasons outside the backend (same on backend="numpy" here)
EXPECTED = {
    "tensornetwork/tests/split_node_test.py": [],
    "tensornetwork/tests/network_operations_test.py": [],
    "tensornetwork/tests/ncon_interface_test.py": [
        r"test_infinite_loop",           # error text embeds repr(np.int64) -- NumPy 2 prints np.int64(3)
    ],
    "tensornetwork/contractors/opt_einsum_paths/path_contractors_node_test.py": [],
    "tensornetwork/tests/tensornetwork_test.py": [],
    "tensornetwork/tests/network_test.py": [],
    "tensornetwork/tests/network_components_free_test.py": [
        r"save|load",                    # HDF5: h5py is not in the image (inert stub)
    ],
    "tensornetwork/tests/tensor_test.py": [
        r"test_init_tensor_from_backend_array",   # the test's own if/elif over the four backend names
    ],
    "tensornetwork/linalg/tests/test_operations.py": [
        r"invalid_backend|test_kron_raises",      # need a second installed backend (tensorflow)
    ],
    "tensornetwork/linalg/tests/test_linalg.py": [],
    "tensornetwork/linalg/tests/initialization_test.py": [],
    "tensornetwork/linalg/tests/node_linalg_test.py": [],
}


def run_file(ref, relpath, backends="hip", emulated=False):
  env = dict(os.environ)
  env["PYTHONDONTWRITEBYTECODE"] = "1"        # nothing is written next to the reference's files
  if emulated:                                # tests/test_reference_dropin_cpu.py: the NumPy emulation of the C ABI
    env["TNH_REF_EMULATED"] = "1"
  env["PYTHONPATH"] = os.pathsep.join([
      os.path.join(REPO, "tests", "golden", "_stubs"), os.path.join(REPO, "tools", "reference_dropin", "_stubs"),
      ref, REPO, os.path.join(REPO, "tools", "reference_dropin"), env.get("PYTHONPATH", "")])
  env["TNH_REF_BACKENDS"] = backends
  cmd = [sys.executable, "-m", "pytest", "--noconftest", "-p", "tnh_ref_plugin", "-p", "no:cacheprovider", "-q",
         "-o", "addopts=", "--timeout=300", "-rfE", relpath]
  proc = subprocess.run(cmd, cwd=ref, env=env, capture_output=True, text=True, timeout=1500, check=False)
  out = proc.stdout + proc.stderr
  failed = re.findall(r"^(?:FAILED|ERROR) (\S+)", out, flags=re.M)
  tail = out.strip().splitlines()[-1] if out.strip() else ""
  m = re.search(r"(\d+) passed", tail)
  return failed, (int(m.group(1)) if m else 0), out


@pytest.mark.parametrize("relpath", sorted(EXPECTED))
def test_reference_test_file_on_hip(hip, relpath):  # pylint: disable=unused-argument
  ref = reference_dir()
  if ref is None:
    pytest.skip("no copy of google/TensorNetwork reachable (set TN_REFERENCE_DIR or ship _reference_scratch/)")
  failed, passed, out = run_file(ref, relpath)
  unexpected = [f for f in failed if not any(re.search(p, f) for p in EXPECTED[relpath])]
  assert passed > 0, out[-3000:]
  assert not unexpected, "\n".join(unexpected) + "\n" + out[-6000:]
