"""GPU parity for the decomposition / dense-linalg rows: HipBackend.qr / rq /
eigh / inv / expm and the host layer's split_node_qr / split_node_rq against the reference's own
outputs (tests/golden/golden_linalg.npz) and against the oracle at larger sizes."""
import numpy as np
import pytest

import tensornetwork_amd as ta
from oracle import numpy_oracle as orc
import cases as C

pytestmark = pytest.mark.gpu


def test_qr_rq_golden(hip, golden_linalg):
  for case in golden_linalg.cases["qr"]:
    x = golden_linalg[case["x"]]
    m = np.asarray(x).reshape(int(np.prod(x.shape[:case["pivot"]])), -1)
    full_rank = np.linalg.matrix_rank(m) == min(m.shape) or not np.any(m)
    C.check_qr_case(hip, golden_linalg, case, tight=bool(full_rank))


def test_split_node_qr_rq_golden(hip, golden_linalg):
  for case in golden_linalg.cases["split_qr"]:
    C.check_split_qr_case(hip, golden_linalg, case)


@pytest.mark.parametrize("dtype,otol,rtol", [(np.float32, 2e-5, 2e-5), (np.float64, 1e-12, 1e-12)])
@pytest.mark.parametrize("shape", [(1024, 1024), (4096, 256), (300, 1000), (2000, 33), (1, 7), (7, 1)])
def test_qr_large_vs_oracle(hip, dtype, otol, rtol, shape):
  """Blocked Householder path (several 32-column panels, trailing GEMM updates): R element-wise
  against np.linalg.qr (same reflector convention), Q orthonormal, Q R = A."""
  rng = np.random.default_rng(shape[0] + shape[1])
  x = rng.standard_normal(shape).astype(dtype)
  q, r = hip.qr(hip.convert_to_tensor(x), 1, False)
  q, r = np.asarray(q), np.asarray(r)
  qo, ro = orc.qr(x, 1, False)
  k = min(shape)
  assert q.shape == (shape[0], k) and r.shape == (k, shape[1]) and q.dtype == dtype
  np.testing.assert_allclose(q.T @ q, np.eye(k), atol=otol * 20)
  scale = np.abs(ro).max()
  np.testing.assert_allclose(q.astype(np.float64) @ r.astype(np.float64), x, atol=rtol * scale * 20)
  np.testing.assert_allclose(r, ro, atol=rtol * scale * 50)
  assert np.array_equal(r, np.triu(r))


def test_qr_bf16_and_errors(hip):
  rng = np.random.default_rng(3)
  x = orc.round_bf16(rng.standard_normal((40, 12)))
  q, r = hip.qr(hip.to_bfloat16(x), 1, True)
  assert q.dtype == np.float32 or str(q.dtype) == "bfloat16"
  qh, rh = np.asarray(q).astype(np.float64), np.asarray(r).astype(np.float64)
  np.testing.assert_allclose(qh @ rh, x, atol=0.05)
  assert np.all(np.diagonal(rh) >= 0)
  with pytest.raises(NotImplementedError):
    hip.qr(hip.convert_to_tensor((x + 1j * x).astype(np.complex64)), 1, False)


def test_eigh_inv_expm_golden(hip, golden_linalg):
  for case in golden_linalg.cases["linalg"]:
    C.check_linalg_case(hip, golden_linalg, case)


@pytest.mark.parametrize("dtype,tol", [(np.float32, 3e-5), (np.float64, 1e-12)])
def test_eigh_large_vs_oracle(hip, dtype, tol):
  rng = np.random.default_rng(8)
  n = 700
  a = rng.standard_normal((n, n))
  h = ((a + a.T) / 2).astype(dtype)
  w, v = hip.eigh(hip.convert_to_tensor(h))
  w, v = np.asarray(w), np.asarray(v)
  wo, _ = orc.eigh(h.astype(np.float64))
  nrm = np.abs(wo).max()
  np.testing.assert_allclose(w, wo, atol=tol * nrm * 5)
  np.testing.assert_allclose(v.T @ v, np.eye(n), atol=tol * 60)
  np.testing.assert_allclose(h @ v, v * w, atol=tol * nrm * 60)
  assert np.all(np.diff(w) >= -tol * nrm)


def test_linalg_errors(hip):
  # numpy_backend.py:555-557, 590-596
  with pytest.raises(ValueError, match="Only matrices are supported"):
    hip.inv(hip.ones((2, 2, 2), dtype=np.float32))
  with pytest.raises(ValueError, match="Only matrices are supported"):
    hip.expm(hip.ones((2, 2, 2), dtype=np.float32))
  with pytest.raises(ValueError, match="only supports N\\*N matrix"):
    hip.expm(hip.ones((2, 3), dtype=np.float32))
