"""Host logic of tensornetwork_amd/krylov.py (Lanczos, thick-restart eigsh, GMRES) driven with the
oracle backend; the GPU suite runs the same functions on HipBackend (tests/test_gpu_linalg.py).
Known answers follow the reference's own tests: numpy_backend_test.py:371-470 (eigsh_lanczos on
dense symmetric matrices vs np.linalg.eigh), :880-918 (gmres on a 2x2 system)."""
import numpy as np
import pytest

from oracle import numpy_oracle as orc
from tensornetwork_amd import krylov


def _sym(n, seed):
  rng = np.random.default_rng(seed)
  a = rng.standard_normal((n, n))
  return (a + a.T) / 2


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_eigsh_lanczos_lowest_pair(dtype):
  be = orc.OracleBackend()
  n = 10
  h = _sym(n, 1).astype(dtype)
  init = np.random.default_rng(2).standard_normal(n).astype(dtype)

  def mv(x, mat):
    return mat @ x

  eta, vecs = krylov.eigsh_lanczos(be, mv, [h], init, num_krylov_vecs=n)
  w, v = np.linalg.eigh(h.astype(np.float64))
  tol = 1e-4 if dtype == np.float32 else 1e-10
  np.testing.assert_allclose(eta[0], w[0], atol=tol * 10)
  v0 = vecs[0] / np.sign(vecs[0][np.argmax(np.abs(v[:, 0]))]) * np.sign(v[np.argmax(np.abs(v[:, 0])), 0])
  np.testing.assert_allclose(v0, v[:, 0], atol=tol * 100)


def test_eigsh_lanczos_reorthogonalize_several_and_shape_dtype_init():
  be = orc.OracleBackend()
  n = 50
  h = _sym(n, 3)
  np.random.seed(10)
  eta, vecs = krylov.eigsh_lanczos(be, lambda x, m: m @ x, [h], shape=(n,), dtype=np.float64,
                                   num_krylov_vecs=n, numeig=3, reorthogonalize=True)
  w = np.linalg.eigvalsh(h)
  np.testing.assert_allclose(eta, w[:3], atol=1e-8)
  for e, v in zip(eta, vecs):
    np.testing.assert_allclose(h @ v, e * v, atol=1e-6)


def test_eigsh_lanczos_errors():
  be = orc.OracleBackend()
  mv = lambda x: x
  with pytest.raises(ValueError, match="`num_krylov_vecs` >= `numeig` required!"):
    krylov.eigsh_lanczos(be, mv, numeig=10, num_krylov_vecs=9, initial_state=np.ones(3))
  with pytest.raises(ValueError, match="Got numeig = 2 > 1 and `reorthogonalize = False`"):
    krylov.eigsh_lanczos(be, mv, numeig=2, reorthogonalize=False, initial_state=np.ones(3))
  with pytest.raises(ValueError, match="if no `initial_state` is passed, then `shape` and"):
    krylov.eigsh_lanczos(be, mv, shape=(10,), dtype=None)
  with pytest.raises(TypeError, match="Expected a backend tensor"):
    krylov.eigsh_lanczos(be, mv, initial_state=[1, 2, 3])


@pytest.mark.parametrize("which", ["LA", "SA", "LM"])
def test_eigsh_thick_restart(which):
  be = orc.OracleBackend()
  n = 120
  h = _sym(n, 4)
  init = np.random.default_rng(5).standard_normal(n)
  eta, vecs = krylov.eigsh(be, lambda x, m: m @ x, [h], init, num_krylov_vecs=24, numeig=4, which=which,
                           tol=1e-10)
  w = np.linalg.eigvalsh(h)
  want = {"LA": w[::-1][:4], "SA": w[:4], "LM": w[np.argsort(-np.abs(w))[:4]]}[which]
  np.testing.assert_allclose(eta, want, atol=1e-8)
  for e, v in zip(eta, vecs):
    np.testing.assert_allclose(h @ v, e * v, atol=1e-6)
  with pytest.raises(ValueError, match="`num_krylov_vecs` > `numeig \\+ 1` required!"):
    krylov.eigsh(be, lambda x: x, initial_state=init, numeig=5, num_krylov_vecs=6)
  with pytest.raises(ValueError, match="which = SI is currently not supported."):
    krylov.eigsh(be, lambda x: x, initial_state=init, which="SI")


def test_gmres_known_answer_and_restarts():
  be = orc.OracleBackend()
  # numpy_backend_test.py:880-897: A = [[1, 1], [3, -4]], b = [3, 2] -> x = [2, 1]
  A = np.array([[1.0, 1.0], [3.0, -4.0]])
  b = np.array([3.0, 2.0])
  x, info = krylov.gmres(be, lambda v, m: m @ v, b, [A], {}, np.zeros(2), 1e-10, 1e-10, 2, 1)
  assert info == 0
  np.testing.assert_allclose(x, [2.0, 1.0], atol=1e-9)
  rng = np.random.default_rng(6)
  n = 80
  M = rng.standard_normal((n, n)) + 3.0 * np.sqrt(n) * np.eye(n)
  rhs = rng.standard_normal((8, 10))                      # arbitrary tensor shape, as the interface allows
  op = lambda v: (M @ v.reshape(-1)).reshape(v.shape)
  x, info = krylov.gmres(be, op, rhs, [], {}, np.zeros(n), 1e-9, 1e-9, 15, 20)
  assert info == 0 and x.shape == rhs.shape
  np.testing.assert_allclose(op(x), rhs, atol=1e-7)
  # not converged within the budget -> info = number of restarts
  x, info = krylov.gmres(be, op, rhs, [], {}, np.zeros(n), 1e-14, 1e-14, 2, 1)
  assert info == 1


def _nonsym(n, seed, cplx=False):
  rng = np.random.default_rng(seed)
  a = rng.standard_normal((n, n))
  if cplx:
    a = a + 1j * rng.standard_normal((n, n))
  return a


def _check_eigs(mat, eta, vecs, which, numeig, atol):
  w = np.linalg.eigvals(mat)
  key = {'LM': -np.abs(w), 'SM': np.abs(w), 'LR': -w.real, 'SR': w.real}[which]
  want = w[np.argsort(key)][:numeig]
  assert eta.dtype == np.complex128 and len(vecs) == numeig
  # conjugate pairs tie in every `which` ordering: compare as sets of nearest matches
  for e in eta:
    assert np.min(np.abs(want - e)) < atol or np.min(np.abs(w - e)) < atol
  key_got = {'LM': -np.abs(eta), 'SM': np.abs(eta), 'LR': -eta.real, 'SR': eta.real}[which]
  np.testing.assert_allclose(np.sort(key_got), np.sort(key[np.argsort(key)][:numeig]), atol=atol)
  for e, v in zip(eta, vecs):
    assert np.iscomplexobj(v)
    np.testing.assert_allclose(np.linalg.norm(v), 1.0, atol=1e-10)
    np.testing.assert_allclose(mat @ v.reshape(-1), e * v.reshape(-1), atol=atol * 10)


@pytest.mark.parametrize("which", ["LM", "LR", "SR"])
@pytest.mark.parametrize("cplx", [False, True])
def test_eigs_krylov_schur_restarts(which, cplx):
  """numpy_backend_test.py:313-370 checks eigs against np.linalg.eig on a dense matrix; same here,
  with ncv << n so the Krylov-Schur restart (and, for real input, conjugate-pair handling) is exercised."""
  be = orc.OracleBackend()
  n = 60
  mat = _nonsym(n, 5, cplx)
  init = np.random.default_rng(6).standard_normal(n).astype(mat.dtype)
  calls = []

  def mv(x, m):
    calls.append(1)
    assert x.dtype == mat.dtype     # a real operator only ever sees real vectors
    return m @ x

  eta, vecs = krylov.eigs(be, mv, [mat], init, num_krylov_vecs=20, numeig=3, which=which, tol=1e-10)
  assert len(calls) > 20            # restarted at least once
  _check_eigs(mat, eta, vecs, which, 3, 1e-7)


def test_eigs_full_space_tensor_shaped_and_sm():
  be = orc.OracleBackend()
  n = 12
  mat = _nonsym(n, 8)
  np.random.seed(3)
  eta, vecs = krylov.eigs(be, lambda x, m: (m @ x.reshape(-1)).reshape(3, 4), [mat], shape=(3, 4),
                          dtype=np.float64, num_krylov_vecs=50, numeig=4, which="SM")
  assert vecs[0].shape == (3, 4)
  _check_eigs(mat, eta, vecs, "SM", 4, 1e-8)


def test_eigs_errors():
  be = orc.OracleBackend()
  mv = lambda x: x
  with pytest.raises(ValueError, match="which = LI is currently not supported."):
    krylov.eigs(be, mv, initial_state=np.ones(3), which="LI")
  with pytest.raises(ValueError, match="which = SI is currently not supported."):
    krylov.eigs(be, mv, initial_state=np.ones(3), which="SI")
  with pytest.raises(ValueError, match="`num_krylov_vecs` > `numeig \\+ 1` required!"):
    krylov.eigs(be, mv, numeig=3, num_krylov_vecs=4, initial_state=np.ones(3))
  with pytest.raises(ValueError, match="if no `initial_state` is passed, then `shape` and"):
    krylov.eigs(be, mv, shape=(10,), dtype=None)
  with pytest.raises(TypeError, match="Expected a backend tensor"):
    krylov.eigs(be, mv, initial_state=[1, 2, 3])


@pytest.mark.parametrize("which", ["SA", "LA"])
def test_eigsh_complex_hermitian(which):
  be = orc.OracleBackend()
  n = 40
  a = _nonsym(n, 9, True)
  h = (a + a.conj().T) / 2
  init = _nonsym(n, 10, True)[0]
  eta, vecs = krylov.eigsh(be, lambda x, m: m @ x, [h], init, num_krylov_vecs=15, numeig=3, which=which, tol=1e-10)
  w = np.linalg.eigvalsh(h)
  want = w[:3] if which == "SA" else w[::-1][:3]
  np.testing.assert_allclose(eta, want, atol=1e-8)
  for e, v in zip(eta, vecs):
    np.testing.assert_allclose(h @ v, e * v, atol=1e-6)


@pytest.mark.parametrize("reorth,numeig", [(False, 1), (True, 3)])
@pytest.mark.parametrize("ndiag", [3, 20])
def test_deferred_lanczos_matches_immediate(reorth, numeig, ndiag):
  """The deferred-readback variant returns what eigsh_lanczos returns (same coefficients, read later)."""
  be = orc.OracleBackend()
  n = 40
  h = _sym(n, 21)
  init = np.random.default_rng(22).standard_normal(n)
  mv = lambda x, m: m @ x
  kw = dict(num_krylov_vecs=25, numeig=numeig, reorthogonalize=reorth, ndiag=ndiag, tol=1e-12)
  e1, v1 = krylov.eigsh_lanczos(be, mv, [h], init, **kw)
  e2, v2 = krylov.eigsh_lanczos_deferred(be, mv, [h], init, **kw)
  np.testing.assert_allclose(e2, e1, rtol=1e-10, atol=1e-10)
  for a, b in zip(v1, v2):
    np.testing.assert_allclose(np.abs(a @ b), 1.0, atol=1e-8)


def test_deferred_lanczos_invariant_subspace_and_errors():
  be = orc.OracleBackend()
  # the start vector spans a 2-dimensional invariant subspace: the third Krylov vector has norm ~ 0
  h = np.diag([1.0, 2.0, 3.0, 4.0, 5.0])
  init = np.array([1.0, 1.0, 0.0, 0.0, 0.0])
  e1, _ = krylov.eigsh_lanczos(be, lambda x: h @ x, initial_state=init, num_krylov_vecs=5, delta=1e-8)
  e2, v2 = krylov.eigsh_lanczos_deferred(be, lambda x: h @ x, initial_state=init, num_krylov_vecs=5, delta=1e-8)
  np.testing.assert_allclose(e2, e1, atol=1e-12)
  np.testing.assert_allclose(e2[0], 1.0, atol=1e-12)
  assert np.all(np.isfinite(v2[0]))
  with pytest.raises(ValueError, match="`num_krylov_vecs` >= `numeig` required!"):
    krylov.eigsh_lanczos_deferred(be, lambda x: x, numeig=10, num_krylov_vecs=9, initial_state=np.ones(3))
  with pytest.raises(TypeError, match="Expected a backend tensor"):
    krylov.eigsh_lanczos_deferred(be, lambda x: x, initial_state=[1, 2, 3])


@pytest.mark.parametrize("deferred", [False, True])
def test_eigsh_lanczos_complex_hermitian_keeps_complex_ritz_coefficients(deferred):
  """for a complex Hermitian operator the eigenvectors of the projected matrix are complex;
  building the Ritz vector with float(u[i, j]) dropped their imaginary parts."""
  import warnings
  be = orc.OracleBackend()
  rng = np.random.default_rng(5)
  n = 12
  a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
  h = (a + a.conj().T) / 2
  init = rng.standard_normal(n) + 1j * rng.standard_normal(n)
  fun = krylov.eigsh_lanczos_deferred if deferred else krylov.eigsh_lanczos
  with warnings.catch_warnings():
    warnings.simplefilter("error")          # a ComplexWarning would mean an imaginary part was discarded
    eta, vecs = fun(be, lambda x, m: m @ x, [h], init, num_krylov_vecs=n, reorthogonalize=True)
  w = np.linalg.eigvalsh(h)
  np.testing.assert_allclose(eta[0], w[0], atol=1e-9)
  v = np.asarray(vecs[0])
  np.testing.assert_allclose(h @ v, w[0] * v, atol=1e-7)
