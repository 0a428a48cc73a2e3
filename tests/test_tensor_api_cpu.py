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


def test_node_linalg_helpers():
  """linalg/tests/node_linalg_test.py: initialisers returning Nodes, conj / transpose / norm / kron."""
  import tensornetwork_amd as ta
  from tensornetwork_amd import node_linalg as nl
  be = OracleBackend()
  e = nl.eye(3, M=4, name="I", axis_names=["r", "c"], backend=be)
  assert isinstance(e, ta.Node) and e.shape == (3, 4) and e.name == "I" and e.axis_names == ["r", "c"]
  np.testing.assert_array_equal(np.asarray(e.tensor), np.eye(3, 4))
  assert nl.zeros((2, 3), dtype=np.float32, backend=be).dtype == np.float32
  assert float(np.asarray(nl.ones((2, 2), backend=be).tensor).sum()) == 4.0
  a, b = nl.randn((2, 3), seed=7, backend=be), nl.randn((2, 3), seed=7, backend=be)
  np.testing.assert_array_equal(np.asarray(a.tensor), np.asarray(b.tensor))
  u = np.asarray(nl.random_uniform((40,), boundaries=(2.0, 3.0), seed=1, backend=be).tensor)
  assert u.min() >= 2.0 and u.max() <= 3.0
  z = np.random.default_rng(0).standard_normal((2, 3, 4)) + 1j * np.random.default_rng(1).standard_normal((2, 3, 4))
  n = ta.Node(z, name="z", axis_names=["a", "b", "c"], backend=be)
  c = nl.conj(n)
  assert c.name == "conj(z)" and c.axis_names == ["a", "b", "c"] and all(x.is_dangling() for x in c.edges)
  np.testing.assert_allclose(np.asarray(c.tensor), z.conj())
  t = nl.transpose(n, ["c", 0, "b"], name="t")
  assert t.shape == (4, 2, 3) and t.axis_names == ["c", "a", "b"] and n.shape == (2, 3, 4)
  np.testing.assert_allclose(np.asarray(t.tensor), z.transpose(2, 0, 1))
  np.testing.assert_allclose(np.asarray(nl.norm(n)), np.linalg.norm(z))
  rng = np.random.default_rng(2)
  x, y, w = rng.standard_normal((2, 3)), rng.standard_normal((2, 3, 4, 5)), rng.standard_normal((3, 2))
  k = nl.kron([ta.Node(x, backend=be), ta.Node(y, backend=be), ta.Node(w, backend=be)])
  assert k.shape == (2, 2, 3, 3, 3, 4, 5, 2)
  np.testing.assert_allclose(np.asarray(k.tensor), np.einsum("ab,cdef,gh->acdgbefh", x, y, w))
  with pytest.raises(ValueError, match="even order"):
    nl.kron([ta.Node(np.ones((2, 2, 2)), backend=be)])
  with pytest.raises(AttributeError):
    nl.conj(3)
