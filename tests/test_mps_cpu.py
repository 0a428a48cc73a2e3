"""Host logic of tensornetwork_amd/mps.py (FiniteMPS gauge moves, MPOs, one-/two-site DMRG) on the
oracle backend; known answers follow matrixproductstates/dmrg_test.py:160-191 (XXZ ground energy vs
exact diagonalisation) and finite_mps_test.py (canonical form checks)."""
import numpy as np
import pytest

from oracle import numpy_oracle as orc
from tensornetwork_amd import mps as tmps
import cases
from cases import xxz_dense


def test_xxz_mpo_matches_kronecker_hamiltonian():
  be = orc.OracleBackend()
  n = 5
  mpo = tmps.xxz_mpo(be, [1.0] * (n - 1), [0.7] * (n - 1), [0.3] * n)
  np.testing.assert_allclose(tmps.mpo_to_dense([np.asarray(w) for w in mpo]), xxz_dense(n, 1.0, 0.7, 0.3), atol=1e-12)


def test_mps_canonical_form_and_position():
  be = orc.OracleBackend()
  state = tmps.FiniteMPS.random([2] * 8, [6] * 7, np.float64, be, seed=3)
  assert state.center_position == 0
  for n in range(1, 8):
    assert state.check_orthonormality("right", n) < 1e-12
  state.position(5)
  for n in range(5):
    assert state.check_orthonormality("left", n) < 1e-12
  for n in range(6, 8):
    assert state.check_orthonormality("right", n) < 1e-12
  np.testing.assert_allclose(np.linalg.norm(state.tensors[5]), 1.0, atol=1e-12)
  # truncating move keeps the requested bond dimension
  state.position(0, D=3)
  assert max(state.bond_dimensions) <= 6 and state.bond_dimensions[1:6] == [2, 3, 3, 3, 3]
  with pytest.raises(ValueError):
    state.position(8)


@pytest.mark.parametrize("n", [4, 6, 7])
def test_dmrg_ground_energy_vs_exact(n):
  # dmrg_test.py:160-191
  be = orc.OracleBackend()
  eta = np.linalg.eigvalsh(xxz_dense(n, 1.0, 1.0, 0.0))
  mpo = tmps.xxz_mpo(be, np.ones(n - 1), np.ones(n - 1), np.zeros(n))
  state = tmps.FiniteMPS.random([2] * n, [32] * (n - 1), np.float64, be, seed=16)
  e1 = tmps.FiniteDMRG(state, mpo).run_one_site(num_sweeps=4, num_krylov_vecs=10)
  np.testing.assert_allclose(e1, eta[0], atol=1e-7)
  state = tmps.FiniteMPS.random([2] * n, [32] * (n - 1), np.float64, be, seed=17)
  dmrg = tmps.FiniteDMRG(state, mpo)
  e2 = dmrg.run_two_site(max_bond_dim=32, num_sweeps=4, num_krylov_vecs=10)
  np.testing.assert_allclose(e2, eta[0], atol=1e-7)
  np.testing.assert_allclose(dmrg.run_two_site(max_bond_dim=32, num_sweeps=0), e2, atol=1e-7)
  np.testing.assert_allclose(dmrg.compute_energy(), eta[0], atol=1e-7)


def test_two_site_gate_and_local_measurement():
  be = orc.OracleBackend()
  n = 6
  state = tmps.FiniteMPS.random([2] * n, [8] * (n - 1), np.float64, be, seed=5)
  sz = np.diag([-0.5, 0.5])
  before = state.measure_local_operator([sz] * n, range(n))
  # identity gate changes nothing; SWAP exchanges the two local expectation values
  state.position(2)
  ident = np.eye(4).reshape(2, 2, 2, 2)
  state.apply_two_site_gate(ident, 2, 3)
  np.testing.assert_allclose(state.measure_local_operator([sz] * n, range(n)), before, atol=1e-12)
  swap = np.zeros((2, 2, 2, 2))
  for a in range(2):
    for b in range(2):
      swap[b, a, a, b] = 1.0
  state.position(2)
  state.apply_two_site_gate(swap, 2, 3)
  after = state.measure_local_operator([sz] * n, range(n))
  np.testing.assert_allclose([after[3], after[2]], [before[2], before[3]], atol=1e-10)


def test_canonicalize_long_chain_float32_and_norm_bookkeeping():
  """A random 32-site state has a norm beyond float32: canonicalize normalises every step and keeps the
  norm on the host; normalize=False puts it back on the centre (small chain, exact check)."""
  be = orc.OracleBackend()
  state = tmps.FiniteMPS.random([2] * 32, [16] * 31, np.float32, be, seed=1)
  assert np.isfinite(np.asarray(state.tensors[0])).all()
  np.testing.assert_allclose(np.linalg.norm(state.tensors[0]), 1.0, atol=1e-5)
  rng = np.random.default_rng(0)
  ts = [rng.standard_normal((1, 2, 3)), rng.standard_normal((3, 2, 3)), rng.standard_normal((3, 2, 1))]
  full = np.einsum("aib,bjc,ckd->ijk", *ts)
  st = tmps.FiniteMPS(ts, be, center_position=1, canonicalize=False)
  z = st.canonicalize(normalize=False)
  np.testing.assert_allclose(z, np.linalg.norm(full), rtol=1e-12)
  np.testing.assert_allclose(np.einsum("aib,bjc,ckd->ijk", *st.tensors), full, atol=1e-12)


@pytest.mark.parametrize("tag", cases.MPS_GOLDEN_TAGS)
def test_mps_measurements_match_reference_golden(tag):
  """left/right_envs, transfer operators, local operators, two-body correlators, one-site gates:
  numbers recorded from the reference's FiniteMPS (tests/golden/make_golden_mps.py)."""
  cases.check_mps_golden_case(orc.OracleBackend(), cases.load_mps_golden(), tag, 1e-12)


def test_mps_measurement_errors():
  be = orc.OracleBackend()
  state = tmps.FiniteMPS.random([2] * 5, [3] * 4, np.float64, be, seed=1)
  with pytest.raises(ValueError, match="have to be <= N"):
    state.left_envs([6])
  with pytest.raises(ValueError, match="have to be positive"):
    state.left_envs([-1])
  with pytest.raises(ValueError, match="have to be < N"):
    state.right_envs([5])
  with pytest.raises(ValueError, match="have to be >= -1"):
    state.right_envs([-2])
  with pytest.raises(ValueError, match="unknown value up for direction"):
    state.apply_transfer_operator(1, "up", np.eye(2))
  with pytest.raises(ValueError, match="len\\(ops\\) has to be len\\(sites\\)"):
    state.measure_local_operator([np.eye(2)], [0, 1])
  with pytest.raises(ValueError, match="rank of gate is 3"):
    state.apply_one_site_gate(np.zeros((2, 2, 2)), 0)
  with pytest.raises(ValueError, match="is not between 0 <= site < N=5"):
    state.apply_one_site_gate(np.eye(2), 5)
  with pytest.raises(IndexError):
    state.get_tensor(5)
  with pytest.raises(ValueError):
    state.get_tensor(-1)
  with pytest.raises(IndexError):
    state.bond_dimension(6)
  with pytest.raises(ValueError, match="Site site1 out of range"):
    state.measure_two_body_correlator(np.eye(2), np.eye(2), -1, [0])
  with pytest.raises(NotImplementedError):
    state.save("x")


class _KrylovSchurOracle(orc.OracleBackend):
  """Oracle arithmetic, but `eigs` through tensornetwork_amd.krylov (the code HipBackend runs)."""

  def eigs(self, *args, **kwargs):
    from tensornetwork_amd import krylov
    return krylov.eigs(self, *args, **kwargs)


@pytest.mark.parametrize("tag", cases.INFINITE_MPS_GOLDEN_TAGS)
@pytest.mark.parametrize("backend_cls", [orc.OracleBackend, _KrylovSchurOracle])
def test_infinite_mps_matches_reference_golden(tag, backend_cls):
  cases.check_infinite_mps_golden_case(backend_cls(), cases.load_mps_golden(), tag, 1e-12)


def test_infinite_mps_errors_and_trivial_bond():
  be = orc.OracleBackend()
  ts = [np.random.randn(2, 2, 10)] + [np.random.randn(10, 2, 10) for _ in range(8)] + [np.random.randn(10, 2, 1)]
  for pos in (-1, 10):
    with pytest.raises(ValueError):                 # infinite_mps_test.py:47-54
      tmps.InfiniteMPS(ts, be, center_position=pos)
  with pytest.raises(ValueError, match="is different from len\\(d\\) \\+ 1"):
    tmps.InfiniteMPS.random([2, 2], [3, 3], np.float64, be)
  with pytest.raises(ValueError, match="D\\[0\\]=3 != D\\[-1\\]=4"):
    tmps.InfiniteMPS.random([2, 2], [3, 3, 4], np.float64, be)
  imps = tmps.InfiniteMPS.random([2, 2], [3, 3, 3], np.float64, be, seed=2)
  with pytest.raises(NotImplementedError):
    imps.left_envs([0])
  with pytest.raises(NotImplementedError):
    imps.right_envs([0])
  imps.center_position = None
  with pytest.raises(ValueError, match="cannot shift `center_position`"):
    imps.position(1)
  product = tmps.InfiniteMPS.random([3], [1, 1], np.float64, be, seed=4)   # D = 1: no boundary entanglement
  eta, mat = product.transfer_matrix_eigs("left")
  t = np.asarray(product.tensors[0]).reshape(3)
  np.testing.assert_allclose(eta, t @ t, rtol=1e-12)
  assert mat.shape == (1, 1)


def test_mpo_classes_match_reference_tensors():
  """FiniteXXZ / FiniteTFI / FiniteFreeFermion2D tensors equal the reference's element for element
  (goldens from mpo.py:129-387 via make_golden_mps.py)."""
  from tensornetwork_amd import mpo as tmpo
  be = orc.OracleBackend()
  g = cases.load_mps_golden()
  jz, jxy, bz = g["mpo_params"][0][:4], g["mpo_params"][1][:4], g["mpo_params"][2]
  models = {"xxz": tmpo.FiniteXXZ(jz, jxy, bz, np.float64, backend=be),
            "tfi": tmpo.FiniteTFI(jz, bz, np.complex128, backend=be)}
  for n1, n2 in ((2, 2), (3, 2), (1, 4)):
    models[f"ff{n1}x{n2}"] = tmpo.FiniteFreeFermion2D(-1.0, -0.7, 0.3, n1, n2, np.float64, backend=be)
  for key, model in models.items():
    assert model.bond_dimensions[0] == 1 and model.bond_dimensions[-1] == 1
    for k, t in enumerate(model):
      ref = g[f"mpo_{key}_{k}"]
      assert t.dtype == ref.dtype and np.array_equal(np.asarray(t), ref), (key, k)
    assert f"mpo_{key}_{len(model)}" not in g
  assert models["xxz"].name == "XXZ_MPO" and models["tfi"].name == "TFI_MPO"


def test_mpo_containers():
  # mpo_test.py:33-92
  from tensornetwork_amd import mpo as tmpo
  be = orc.OracleBackend()
  ts = [be.randn((1, 5, 2, 2), dtype=np.float64), be.randn((5, 5, 2, 2), dtype=np.float64),
        be.randn((5, 1, 2, 2), dtype=np.float64)]
  m = tmpo.BaseMPO(ts, backend=be, name="test")
  assert m.backend is be and m.dtype == np.float64 and m.bond_dimensions == [1, 5, 5, 1] and len(m) == 3
  bad = ts[:2] + [be.randn((5, 1, 2, 2), dtype=np.float32)]
  with pytest.raises(TypeError):
    tmpo.BaseMPO(bad, backend=be)
  empty = tmpo.BaseMPO([], backend=be)
  empty.tensors = bad
  with pytest.raises(TypeError):
    empty.dtype  # pylint: disable=pointless-statement
  with pytest.raises(ValueError):
    tmpo.FiniteMPO([np.random.rand(2, 5, 2, 2), np.random.rand(5, 1, 2, 2)], backend=be)
  with pytest.raises(ValueError):
    tmpo.FiniteMPO([np.random.rand(1, 5, 2, 2), np.random.rand(5, 2, 2, 2)], backend=be)
  with pytest.raises(ValueError):
    tmpo.InfiniteMPO([np.random.rand(2, 5, 2, 2), np.random.rand(5, 3, 2, 2)], backend=be)
  pair = [np.random.rand(5, 5, 2, 2), np.random.rand(5, 5, 2, 2)]
  inf = tmpo.InfiniteMPO(pair, backend=be)
  inf.roll(1)
  np.testing.assert_array_equal(inf.tensors[0], pair[1])
  inf.roll(1)
  np.testing.assert_array_equal(inf.tensors[0], pair[0])


def free_fermion_ground_energy(n1, n2, t, v):
  """Exact: fill the negative levels of the single-particle hopping matrix (mpo_test.py:95-117)."""
  n = n1 * n2
  tij = np.zeros((n, n))
  for s in range(n):
    col, row = divmod(s, n1)                         # snake: n1 sites per column
    if row < n1 - 1:
      tij[s, s + 1] = tij[s + 1, s] = t
    if col < n2 - 1:
      tij[s, s + n1] = tij[s + n1, s] = t
  tij += v * np.eye(n)
  return min(np.cumsum(np.linalg.eigvalsh(tij)))


@pytest.mark.parametrize("n1,n2,D", [(2, 2, 4), (2, 4, 16)])
def test_free_fermion_2d_dmrg_ground_energy(n1, n2, D):
  # mpo_test.py:94-128: one-site DMRG on the snaked 2-D free-fermion MPO vs the filled Fermi sea
  from tensornetwork_amd import mpo as tmpo
  be = orc.OracleBackend()
  model = tmpo.FiniteFreeFermion2D(-1.0, -1.0, -1.0, n1, n2, np.float64, backend=be)
  state = tmps.FiniteMPS.random([2] * (n1 * n2), [D] * (n1 * n2 - 1), np.float64, be, seed=5)
  energy = tmps.FiniteDMRG(state, model).run_one_site(num_sweeps=6, precision=1e-10)
  np.testing.assert_allclose(energy, free_fermion_ground_energy(n1, n2, -1.0, -1.0), rtol=1e-6)


@pytest.mark.parametrize("n,field", [(6, 1.0), (8, 0.5)])
def test_tfi_dmrg_and_correlators_vs_exact_diagonalisation(n, field):
  """FiniteTFI through two-site DMRG: ground energy, <Z_i> and <X_0 X_j> against the dense Hamiltonian's
  ground state (dmrg_test.py / finite_mps_test.py pattern: MPS measurements vs exact vectors)."""
  from tensornetwork_amd import mpo as tmpo
  be = orc.OracleBackend()
  sx, sz, one = np.array([[0.0, 1.0], [1.0, 0.0]]), np.diag([-1.0, 1.0]), np.eye(2)

  def site_op(op, k):
    mats = [one] * n
    mats[k] = op
    out = mats[0]
    for m in mats[1:]:
      out = np.kron(out, m)
    return out

  ham = sum(site_op(sx, k) @ site_op(sx, k + 1) for k in range(n - 1)) + field * sum(site_op(sz, k) for k in range(n))
  w, v = np.linalg.eigh(ham)
  gs = v[:, 0]
  model = tmpo.FiniteTFI(np.ones(n - 1), field * np.ones(n), np.float64, backend=be)
  np.testing.assert_allclose(tmps.mpo_to_dense([np.asarray(t) for t in model]), ham, atol=1e-12)
  state = tmps.FiniteMPS.random([2] * n, [16] * (n - 1), np.float64, be, seed=3)
  energy = tmps.FiniteDMRG(state, model).run_two_site(max_bond_dim=16, num_sweeps=6, precision=1e-10,
                                                      num_krylov_vecs=12)
  np.testing.assert_allclose(energy, w[0], atol=1e-8)
  z_exact = [gs @ site_op(sz, k) @ gs for k in range(n)]
  np.testing.assert_allclose(state.measure_local_operator([sz] * n, range(n)), z_exact, atol=1e-5)
  xx_exact = [gs @ site_op(sx, 0) @ site_op(sx, j) @ gs for j in range(1, n)]
  np.testing.assert_allclose(state.measure_two_body_correlator(sx, sx, 0, range(1, n)), xx_exact, atol=1e-5)
  mid = n // 2
  xx_mid = [gs @ site_op(sx, mid) @ site_op(sx, j) @ gs for j in range(n)]
  np.testing.assert_allclose(state.measure_two_body_correlator(sx, sx, mid, range(n)), xx_mid, atol=1e-5)


@pytest.mark.parametrize("backend_cls", [orc.OracleBackend, _KrylovSchurOracle])
def test_infinite_mps_aklt_known_answer(backend_cls):
  """The AKLT state as a one-site unit cell, scrambled by a random gauge: the transfer matrix has dominant
  eigenvalue 1 and canonicalize() must find the two equal Schmidt values 1/sqrt(2) again."""
  be = backend_cls()
  sp, sm, sz = np.array([[0.0, 1.0], [0.0, 0.0]]), np.array([[0.0, 0.0], [1.0, 0.0]]), np.diag([1.0, -1.0])
  aklt = np.stack([np.sqrt(2 / 3) * sp, -np.sqrt(1 / 3) * sz, -np.sqrt(2 / 3) * sm], axis=1)   # (D, d, D)
  g = np.random.default_rng(5).standard_normal((2, 2)) + 2.0 * np.eye(2)
  scrambled = np.einsum("ab,bsc,cd->asd", np.linalg.inv(g), aklt, g)
  imps = tmps.InfiniteMPS([scrambled], be, center_position=0)
  np.random.seed(2)
  eta, _ = imps.transfer_matrix_eigs("left")
  np.testing.assert_allclose(eta, 1.0, atol=1e-9)
  imps.canonicalize()
  schmidt = np.sort(np.abs(1.0 / np.diag(np.asarray(imps.connector_matrix))))
  np.testing.assert_allclose(schmidt, [np.sqrt(0.5)] * 2, atol=1e-8)
  assert imps.check_orthonormality("l", 0) < 1e-8


def test_dmrg_with_deferred_lanczos_readbacks():
  """FiniteDMRG.deferred_lanczos = True (device-resident Krylov coefficients) reaches the same ground energy."""
  be = orc.OracleBackend()
  n = 6
  eta = np.linalg.eigvalsh(xxz_dense(n, 1.0, 1.0, 0.0))
  mpo = tmps.xxz_mpo(be, np.ones(n - 1), np.ones(n - 1), np.zeros(n))
  state = tmps.FiniteMPS.random([2] * n, [16] * (n - 1), np.float64, be, seed=16)
  dmrg = tmps.FiniteDMRG(state, mpo)
  dmrg.deferred_lanczos = True
  np.testing.assert_allclose(dmrg.run_two_site(max_bond_dim=16, num_sweeps=4, num_krylov_vecs=10), eta[0], atol=1e-7)


def test_one_site_mps_canonicalize_returns_the_true_norm():
  """position(site == center_position) dropped the norm from the bookkeeping."""
  from oracle.numpy_oracle import OracleBackend
  import tensornetwork_amd as ta
  be = OracleBackend()
  t = np.random.default_rng(3).standard_normal((1, 4, 1))
  true = float(np.linalg.norm(t))
  mps = ta.FiniteMPS([t.copy()], center_position=0, canonicalize=False, backend=be)
  assert float(mps.canonicalize(normalize=True)) == pytest.approx(true)
  np.testing.assert_allclose(np.linalg.norm(np.asarray(mps.tensors[0])), 1.0)
  mps = ta.FiniteMPS([t.copy()], center_position=0, canonicalize=False, backend=be)
  assert float(mps.canonicalize(normalize=False)) == pytest.approx(true)
  np.testing.assert_allclose(np.asarray(mps.tensors[0]), t, rtol=1e-12)
