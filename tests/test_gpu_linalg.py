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


@pytest.mark.parametrize("dtype,otol,rtol", [(np.float32, 2e-5, 2e-5), (np.float64, 1e-12, 1e-12)])
@pytest.mark.parametrize("shape", [(1536, 1024), (2048, 512), (1000, 608)])
def test_qr_fused_panels_and_the_five_launch_loop_agree_with_lapack(hip, monkeypatch, dtype, otol, rtol, shape):
  """Round 6 (qr_panels_fast): the leading panels of the 16-wide Householder QR run as look-ahead + fused update /
  raw pass / factor + reduce; TNH_SVDB_FAST=0 keeps the five-launch loop of rounds 3-5.  Both against np.linalg.qr
  (same reflector signs), Q orthonormal, Q R = A."""
  rng = np.random.default_rng(shape[0] * 3 + shape[1])
  x = rng.standard_normal(shape).astype(dtype)
  qo, ro = orc.qr(x, 1, False)
  scale = np.abs(ro).max()
  for fast in ("1", "0"):
    monkeypatch.setenv("TNH_SVDB_FAST", fast)
    q, r = hip.qr(hip.convert_to_tensor(x), 1, False)
    q, r = np.asarray(q), np.asarray(r)
    np.testing.assert_allclose(q.T @ q, np.eye(shape[1]), atol=otol * 20, err_msg=fast)
    np.testing.assert_allclose(q.astype(np.float64) @ r.astype(np.float64), x, atol=rtol * scale * 20, err_msg=fast)
    np.testing.assert_allclose(r, ro, atol=rtol * scale * 50, err_msg=fast)
    assert np.array_equal(r, np.triu(r))


def test_complex_svd_golden(hip, golden_complex):
  """complex64 / complex128 svd (one-sided Jacobi with unitary rotations) vs the reference's outputs."""
  for case in golden_complex.cases["svd"]:
    C.check_svd_case(hip, golden_complex, case)
  # zero matrix: LAPACK's completion (split_node_test.py:22-32 with a complex dtype)
  u, s, vh, _ = hip.svd(hip.zeros((4, 6), dtype=np.complex64), 1)
  np.testing.assert_array_equal(np.asarray(s), 0)
  np.testing.assert_array_equal(np.asarray(u), np.eye(4))
  np.testing.assert_array_equal(np.asarray(vh), np.eye(4, 6))


def test_complex_qr_rq_golden(hip, golden_complex):
  """Complex QR through the real Householder kernels on the interleaved embedding: with
  non_negative_diagonal the factorisation is unique and is compared element-wise; without it only
  Q^H Q = 1, Q R = A and the triangular shape (R comes out in the non-negative-diagonal gauge)."""
  for case in golden_complex.cases["qr"]:
    C.check_qr_case(hip, golden_complex, case, tight=bool(case["nnd"]))
  q, r = hip.qr(hip.zeros((5, 3), dtype=np.complex128), 1, False)
  np.testing.assert_array_equal(np.asarray(q), np.eye(5, 3))
  np.testing.assert_array_equal(np.asarray(r), 0)


def test_complex_eigh_inv_expm_golden(hip, golden_complex):
  for case in golden_complex.cases["linalg"]:
    C.check_linalg_case(hip, golden_complex, case)


def test_complex_split_node_golden(hip, golden_complex):
  from tensornetwork_amd import network
  case = golden_complex.cases["split"][0]
  x = golden_complex[case["x"]]
  a = network.Node(hip.convert_to_tensor(x), backend=hip)
  l, r, tr = network.split_node(a, [a[0], a[1]], [a[2], a[3]], max_singular_values=4)
  C.assert_close(tr, golden_complex[case["trun"]], scale=float(np.abs(x).max()) * 10)
  got = np.tensordot(np.asarray(l.tensor), np.asarray(r.tensor), [[2], [0]])
  ref = np.tensordot(golden_complex[case["left"]], golden_complex[case["right"]], [[2], [0]])
  np.testing.assert_allclose(got, ref, atol=1e-10)


def test_complex_svd_larger_vs_oracle(hip):
  rng = np.random.default_rng(5)
  x = (rng.standard_normal((300, 200)) + 1j * rng.standard_normal((300, 200))).astype(np.complex64)
  u, s, vh, rest = hip.svd(hip.convert_to_tensor(x), 1, max_singular_values=50)
  so = np.linalg.svd(x.astype(np.complex128), compute_uv=False)
  np.testing.assert_allclose(np.real(np.asarray(s)), so[:50], atol=2e-5 * so[0])
  np.testing.assert_allclose(np.real(np.asarray(rest)), so[50:], atol=2e-5 * so[0])
  um, vm = np.asarray(u), np.asarray(vh)
  np.testing.assert_allclose(um.conj().T @ um, np.eye(50), atol=2e-4)
  np.testing.assert_allclose(vm @ vm.conj().T, np.eye(50), atol=2e-4)
  uo, _, vo = np.linalg.svd(x.astype(np.complex128), full_matrices=False)
  ref = (uo[:, :50] * so[:50]) @ vo[:50]
  np.testing.assert_allclose((um * np.asarray(s)) @ vm, ref, atol=2e-3)


def test_qr_bf16_and_errors(hip):
  rng = np.random.default_rng(3)
  x = orc.round_bf16(rng.standard_normal((40, 12)))
  q, r = hip.qr(hip.to_bfloat16(x), 1, True)
  assert q.dtype == np.float32 or str(q.dtype) == "bfloat16"
  qh, rh = np.asarray(q).astype(np.float64), np.asarray(r).astype(np.float64)
  np.testing.assert_allclose(qh @ rh, x, atol=0.05)
  assert np.all(np.diagonal(rh) >= 0)


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


# ------------------------------------------------------------------ Krylov / masks
def _sym(n, seed, dtype):
  rng = np.random.default_rng(seed)
  a = rng.standard_normal((n, n))
  return ((a + a.T) / 2).astype(dtype)


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-4), (np.float64, 1e-9)])
def test_eigsh_lanczos_device_vectors(hip, dtype, tol):
  """numpy_backend_test.py:371-470 on the hip backend: operator = device GEMV through tensordot."""
  n = 64
  h = _sym(n, 1, dtype)
  hd = hip.convert_to_tensor(h)
  init = hip.convert_to_tensor(np.random.default_rng(2).standard_normal(n).astype(dtype))

  def mv(x, mat):
    return hip.tensordot(mat, x, ([1], [0]))

  eta, vecs = hip.eigsh_lanczos(mv, [hd], init, num_krylov_vecs=n, numeig=2, reorthogonalize=True)
  w, v = np.linalg.eigh(h.astype(np.float64))
  np.testing.assert_allclose(eta, w[:2], atol=tol * 10)
  for e, vec in zip(eta, vecs):
    assert isinstance(vec, ta.DeviceTensor) and vec.dtype == dtype
    vh = np.asarray(vec).astype(np.float64)
    np.testing.assert_allclose(h @ vh, e * vh, atol=tol * 100)
  # random initial state from shape / dtype, tensor-shaped vectors
  np.random.seed(3)
  eta, vecs = hip.eigsh_lanczos(lambda x, mat: hip.reshape(hip.tensordot(mat, hip.reshape(x, (n,)), 1), (8, 8)),
                                [hd], shape=(8, 8), dtype=dtype, num_krylov_vecs=n)
  np.testing.assert_allclose(eta[0], w[0], atol=tol * 10)
  assert vecs[0].shape == (8, 8)
  with pytest.raises(TypeError, match="Expected a `DeviceTensor`"):
    hip.eigsh_lanczos(mv, [hd], initial_state=np.ones(n))


def test_eigsh_and_gmres_device_vectors(hip):
  n = 100
  h = _sym(n, 4, np.float64)
  hd = hip.convert_to_tensor(h)
  init = hip.convert_to_tensor(np.random.default_rng(5).standard_normal(n))
  mv = lambda x, mat: hip.tensordot(mat, x, 1)
  eta, vecs = hip.eigsh(mv, [hd], init, num_krylov_vecs=20, numeig=3, which="SA", tol=1e-10)
  w = np.linalg.eigvalsh(h)
  np.testing.assert_allclose(eta, w[:3], atol=1e-8)
  for e, vec in zip(eta, vecs):
    vh = np.asarray(vec)
    np.testing.assert_allclose(h @ vh, e * vh, atol=1e-6)
  # gmres: numpy_backend_test.py:880-897 known answer, then a tensor-shaped system
  A = hip.convert_to_tensor(np.array([[1.0, 1.0], [3.0, -4.0]]))
  b = hip.convert_to_tensor(np.array([3.0, 2.0]))
  x, info = hip.gmres(lambda v, m: hip.tensordot(m, v, 1), b, A_args=[A], tol=1e-10, num_krylov_vectors=2)
  assert info == 0
  np.testing.assert_allclose(np.asarray(x), [2.0, 1.0], atol=1e-9)
  rng = np.random.default_rng(6)
  M = rng.standard_normal((n, n)) + 3.0 * np.sqrt(n) * np.eye(n)
  Md = hip.convert_to_tensor(M)
  rhs = rng.standard_normal((10, 10))
  op = lambda v: hip.reshape(hip.tensordot(Md, hip.reshape(v, (n,)), 1), (10, 10))
  x, info = hip.gmres(op, hip.convert_to_tensor(rhs), tol=1e-9, num_krylov_vectors=15, maxiter=20)
  assert info == 0 and x.shape == (10, 10)
  np.testing.assert_allclose((M @ np.asarray(x).reshape(-1)).reshape(10, 10), rhs, atol=1e-7)
  with pytest.raises(ValueError, match="must match b's"):
    hip.gmres(op, hip.convert_to_tensor(rhs), x0=hip.zeros((5,), dtype=np.float64))
  with pytest.raises(TypeError, match="must match b's"):
    hip.gmres(op, hip.convert_to_tensor(rhs), x0=hip.zeros((10, 10), dtype=np.float32))


@pytest.mark.parametrize("dtype", [np.float64, np.complex128, np.float32])
def test_eigs_device_vectors(hip, dtype):
  """numpy_backend_test.py:313-370 (eigs vs np.linalg.eig) with ncv << n: Krylov-Schur restarts on device vectors."""
  n = 80
  rng = np.random.default_rng(11)
  mat = rng.standard_normal((n, n))
  if np.dtype(dtype).kind == "c":
    mat = mat + 1j * rng.standard_normal((n, n))
  mat = mat.astype(dtype)
  md = hip.convert_to_tensor(mat)
  init = hip.convert_to_tensor(rng.standard_normal(n).astype(dtype))
  seen = []

  def mv(x, m):
    seen.append(x.dtype)
    return hip.tensordot(m, x, 1)

  tol, atol = (1e-5, 2e-3) if dtype == np.float32 else (1e-10, 1e-7)
  eta, vecs = hip.eigs(mv, [md], init, num_krylov_vecs=24, numeig=3, which="LM", tol=tol)
  assert all(d == dtype for d in seen) and len(seen) > 24
  w = np.linalg.eigvals(mat.astype(np.complex128))
  want = w[np.argsort(-np.abs(w))][:3]
  np.testing.assert_allclose(np.sort(np.abs(eta)), np.sort(np.abs(want)), atol=atol)
  m128 = mat.astype(np.complex128)
  for e, vec in zip(eta, vecs):
    assert isinstance(vec, ta.DeviceTensor) and vec.is_complex
    vh = np.asarray(vec).astype(np.complex128)
    np.testing.assert_allclose(m128 @ vh, e * vh, atol=atol * 10)
  with pytest.raises(ValueError, match="which = LI is currently not supported."):
    hip.eigs(mv, [md], init, which="LI")
  with pytest.raises(TypeError, match="Expected a `DeviceTensor`"):
    hip.eigs(mv, [md], initial_state=np.ones(n))


def test_eigsh_complex_hermitian_and_pivot(hip):
  n = 48
  rng = np.random.default_rng(12)
  a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
  h = (a + a.conj().T) / 2
  hd = hip.convert_to_tensor(h)
  init = hip.convert_to_tensor(rng.standard_normal(n) + 1j * rng.standard_normal(n))
  eta, vecs = hip.eigsh(lambda x, m: hip.tensordot(m, x, 1), [hd], init, num_krylov_vecs=16, numeig=2,
                        which="SA", tol=1e-10)
  w = np.linalg.eigvalsh(h)
  np.testing.assert_allclose(eta, w[:2], atol=1e-8)
  for e, vec in zip(eta, vecs):
    vh = np.asarray(vec)
    np.testing.assert_allclose(h @ vh, e * vh, atol=1e-6)
  # pivot (abstract_backend.py:938-962; backend_test / numpy_backend_test pivot cases)
  t = hip.convert_to_tensor(np.arange(1 * 2 * 4 * 5, dtype=np.float64).reshape(1, 2, 4, 5))
  assert hip.pivot(t, 2).shape == (2, 20)
  assert hip.pivot(t).shape == (8, 5)
  np.testing.assert_array_equal(np.asarray(hip.pivot(t, 1)), np.arange(40.0).reshape(1, 40))
  with pytest.raises(ValueError, match="was invalid given ndim=4"):
    hip.pivot(t, 5)
  with pytest.raises(NotImplementedError):
    hip.cholesky(t)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, ta.bfloat16])
def test_compare_and_index_update(hip, dtype):
  """The reference's pattern (infinite_mps.py:237-241): mask = eigvals <= precision;
  index_update(eigvals, mask, 0.0); index_update(1 / eigvals, mask, 0.0) -- inf under the mask
  must not leak (a multiplicative blend would give nan)."""
  x = np.array([[1.0, -2.0, 0.0, 1e-3], [0.5, 0.0, -1e-3, 4.0]], dtype=np.float32)
  d = hip.to_bfloat16(x) if dtype is ta.bfloat16 else hip.convert_to_tensor(x.astype(dtype))
  xr = np.asarray(d).astype(np.float64)
  mask = d <= 1e-2
  np.testing.assert_array_equal(np.asarray(mask), (xr <= 1e-2).astype(np.int32))
  for op, fn in [("<", np.less), (">", np.greater), (">=", np.greater_equal), ("==", np.equal), ("!=", np.not_equal)]:
    np.testing.assert_array_equal(np.asarray(hip.compare(op, d, 0.0)), fn(xr, 0.0).astype(np.int32))
  np.testing.assert_array_equal(np.asarray(d > hip.multiply(d, 0.5)), (xr > 0.5 * xr).astype(np.int32))
  upd = np.asarray(hip.index_update(d, mask, 0.0)).astype(np.float64)
  np.testing.assert_array_equal(upd, np.where(xr <= 1e-2, 0.0, xr))
  inv = hip.divide(1.0, d)
  got = np.asarray(hip.index_update(inv, mask, 0.0)).astype(np.float64)
  assert np.all(np.isfinite(got))
  np.testing.assert_allclose(got, np.where(xr <= 1e-2, 0.0, 1.0 / np.where(xr == 0, 1, xr)), rtol=1e-2)
  # host boolean mask (numpy_backend_test.py:720-727 passes `tensor > 0.1` computed anywhere)
  got = np.asarray(hip.index_update(d, xr > 0.1, 9.0)).astype(np.float64)
  np.testing.assert_array_equal(got, np.where(xr > 0.1, 9.0, xr))


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float64, 1e-12)])
def test_qr_tall_panels_rank_deficient(hip, dtype, tol):
  """Multi-workgroup panel path (m_j > 512) on inputs where reflectors degenerate: zeros (tau = 0,
  Q = I_thin as LAPACK), a rank-1 matrix and duplicated columns -- Q stays orthonormal, Q R = A."""
  rng = np.random.default_rng(11)
  cases = [np.zeros((2000, 40)), np.outer(rng.standard_normal(1500), rng.standard_normal(33)),
           np.repeat(rng.standard_normal((1300, 20)), 2, axis=1)]
  for a in cases:
    a = a.astype(dtype)
    q, r = hip.qr(hip.convert_to_tensor(a), 1, False)
    q, r = np.asarray(q).astype(np.float64), np.asarray(r).astype(np.float64)
    k = min(a.shape)
    scale = max(np.abs(a).max(), 1e-30) * np.sqrt(a.shape[0])
    np.testing.assert_allclose(q.T @ q, np.eye(k), atol=tol * 50)
    np.testing.assert_allclose(q @ r, a, atol=tol * scale * 50)
    assert np.array_equal(r, np.triu(r))
  q, r = hip.qr(hip.convert_to_tensor(np.zeros((2000, 40), dtype=dtype)), 1, False)
  np.testing.assert_array_equal(np.asarray(q), np.eye(2000, 40))
  np.testing.assert_array_equal(np.asarray(r), 0)


def test_tensor_and_functional_api_on_device(hip):
  """`Tensor` operators + tn.linalg-style functional API (tensor.py, linalg.py) on device tensors: the CPU
  suite's checker (tests/cases.py:check_tensor_api) with HipBackend."""
  import cases
  cases.check_tensor_api(hip, 1e-9)


@pytest.mark.parametrize("dtype,stol,otol", [(np.float32, 2e-6, 2e-5), (np.float64, 1e-12, 1e-11)])
@pytest.mark.parametrize("n,kind", [(256, "graded2"), (256, "rank40"), (384, "dmrg_like"), (256, "zero_rows")])
def test_svd_graded_and_rank_deficient_block_path(hip, dtype, stol, otol, n, kind):
  """Block Jacobi on inputs whose singular values span many decades or vanish (what every DMRG split
  hands to svd): all singular values to eps * s_1 against LAPACK, U and Vh orthonormal INCLUDING the
  near-null vectors, reconstruction to eps.  Regression: a rotation test on squared Gram entries underflowed
  in f32 (rows below 1e-10 of the largest read "already orthogonal"), and 40 sweeps were not enough for f64."""
  rng = np.random.default_rng(5)
  u, _ = np.linalg.qr(rng.standard_normal((n, n)))
  v, _ = np.linalg.qr(rng.standard_normal((n, n)))
  if kind == "graded2":
    s = 2.0 ** -np.arange(n)
  elif kind == "rank40":
    s = np.concatenate([np.linspace(1, 0.1, 40), np.zeros(n - 40)])
  elif kind == "zero_rows":
    s = np.concatenate([np.ones(n // 2), np.zeros(n - n // 2)])
  else:
    s = np.exp(-np.arange(n) / 6.0)
  a = (u * s) @ v.T
  if kind == "zero_rows":
    a[::3] = 0.0
  a = a.astype(dtype)
  uu, ss, vv, _ = hip.svd(hip.convert_to_tensor(a), 1)
  a64 = a.astype(np.float64)
  s_ref = np.linalg.svd(a64, compute_uv=False)
  uh, sh, vh = (np.asarray(t).astype(np.float64) for t in (uu, ss, vv))
  assert np.abs(sh - s_ref).max() <= stol * s_ref[0]
  assert np.abs(uh.T @ uh - np.eye(n)).max() <= otol
  assert np.abs(vh @ vh.T - np.eye(n)).max() <= otol
  assert np.abs((uh * sh) @ vh - a64).max() <= 10 * stol * s_ref[0]
