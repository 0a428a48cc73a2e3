"""CPU suite for the decomposition / dense-linalg rows: pins the oracle's
qr / rq / eigh / inv / expm restatements against outputs of the reference itself
(tests/golden/golden_linalg.npz) and drives the host layer's split_node_qr / split_node_rq with
the oracle backend through the same checks the GPU suite applies to HipBackend."""
import numpy as np
import pytest

from oracle import numpy_oracle as orc
import cases as C


def test_oracle_qr_rq_match_reference(golden_linalg):
  be = orc.OracleBackend()
  for case in golden_linalg.cases["qr"]:
    C.check_qr_case(be, golden_linalg, case)


def test_oracle_qr_known_answers():
  # backends/numpy/decompositions_test.py:31-53: q r reconstructs, shapes at pivot_axis
  rng = np.random.default_rng(5)
  x = rng.standard_normal((2, 3, 4, 5))
  q, r = orc.qr(x, 2, False)
  assert q.shape == (2, 3, 6) and r.shape == (6, 4, 5)
  np.testing.assert_allclose(np.tensordot(q, r, ([2], [0])), x, atol=1e-12)
  r2, q2 = orc.rq(x, 2, False)
  assert r2.shape == (2, 3, 6) and q2.shape == (6, 4, 5)
  np.testing.assert_allclose(np.tensordot(r2, q2, ([2], [0])), x, atol=1e-12)
  # non_negative_diagonal (decompositions_test.py:110-130)
  _, r = orc.qr(x, 2, True)
  assert np.all(np.diagonal(r.reshape(6, 20)) >= 0)
  r2, _ = orc.rq(x, 2, True)
  assert np.all(np.diagonal(r2.reshape(6, 6)) >= 0)


def test_host_split_node_qr_rq_with_oracle_backend(golden_linalg):
  be = orc.OracleBackend()
  for case in golden_linalg.cases["split_qr"]:
    C.check_split_qr_case(be, golden_linalg, case)


def test_oracle_eigh_inv_expm_match_reference(golden_linalg):
  be = orc.OracleBackend()
  for case in golden_linalg.cases["linalg"]:
    C.check_linalg_case(be, golden_linalg, case)


def test_inv_expm_errors():
  # numpy_backend.py:555-557, 590-596
  with pytest.raises(ValueError, match="Only matrices are supported"):
    orc.inv(np.ones((2, 2, 2)))
  with pytest.raises(ValueError, match="Only matrices are supported"):
    orc.expm(np.ones((2, 2, 2)))
  with pytest.raises(ValueError, match="only supports N\\*N matrix"):
    orc.expm(np.ones((2, 3)))
