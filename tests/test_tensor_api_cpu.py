"""`Tensor` + functional API (tensor.py / linalg.py) on the oracle backend; the GPU suite runs the same
checker on HipBackend (tests/test_gpu_linalg.py)."""
import numpy as np
import pytest

import cases
from oracle.numpy_oracle import OracleBackend
from tensornetwork_amd import tensor as tt, linalg as tl


def test_tensor_and_functional_api():
  cases.check_tensor_api(OracleBackend(), 1e-10)


def test_backend_mismatch_errors():
  class Other(OracleBackend):
    name = "other"
  a, b = tt.Tensor(np.ones((2, 2)), backend=OracleBackend()), tt.Tensor(np.ones((2, 2)), backend=Other())
  with pytest.raises(ValueError, match="differing backends"):
    tl.tensordot(a, b, 1)
  with pytest.raises(ValueError, match="inconsistent"):
    a + b  # pylint: disable=pointless-statement
  with pytest.raises(ValueError, match="did not agree"):
    a @ b  # pylint: disable=pointless-statement
  with pytest.raises(ValueError, match="must have the same backend"):
    tl.outer(a, b)
  with pytest.raises(ValueError, match="backends must agree"):
    tl.eigsh_lanczos(lambda x: x, backend=Other(), x0=a)
