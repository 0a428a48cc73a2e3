"""MPS / DMRG end to end on the GPU: gauge moves through the device QR / RQ,
two-site splits through the Jacobi SVD, Lanczos on device vectors.  Known answers as in
matrixproductstates/dmrg_test.py:160-191 (ground energy vs exact diagonalisation)."""
import numpy as np
import pytest

import tensornetwork_amd as ta
from tensornetwork_amd import mps as tmps
import cases
from cases import xxz_dense

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-7), (np.float32, 2e-4)])
def test_mps_canonical_form_on_device(hip, dtype, tol):
  state = tmps.FiniteMPS.random([2] * 8, [6] * 7, dtype, hip, seed=3)
  assert isinstance(state.tensors[0], ta.DeviceTensor)
  for n in range(1, 8):
    assert state.check_orthonormality("right", n) < tol
  state.position(5)
  for n in range(5):
    assert state.check_orthonormality("left", n) < tol
  np.testing.assert_allclose(np.linalg.norm(np.asarray(state.tensors[5])), 1.0, atol=tol)
  state.position(0, D=3)
  assert state.bond_dimensions[1:6] == [2, 3, 3, 3, 3]


@pytest.mark.parametrize("n", [4, 7])
def test_dmrg_ground_energy_vs_exact_f64(hip, n):
  eta = np.linalg.eigvalsh(xxz_dense(n, 1.0, 1.0, 0.0))
  mpo = tmps.xxz_mpo(hip, np.ones(n - 1), np.ones(n - 1), np.zeros(n))
  state = tmps.FiniteMPS.random([2] * n, [32] * (n - 1), np.float64, hip, seed=16)
  e1 = tmps.FiniteDMRG(state, mpo).run_one_site(num_sweeps=4, num_krylov_vecs=10)
  np.testing.assert_allclose(e1, eta[0], atol=1e-7)
  state = tmps.FiniteMPS.random([2] * n, [32] * (n - 1), np.float64, hip, seed=17)
  dmrg = tmps.FiniteDMRG(state, mpo)
  e2 = dmrg.run_two_site(max_bond_dim=32, num_sweeps=4, num_krylov_vecs=10)
  np.testing.assert_allclose(e2, eta[0], atol=1e-7)
  np.testing.assert_allclose(dmrg.compute_energy(), eta[0], atol=1e-7)


def test_dmrg_f32_chain_of_12(hip):
  n = 12
  eta = np.linalg.eigvalsh(xxz_dense(n, 1.0, 1.0, 0.0))
  mpo = tmps.xxz_mpo(hip, np.ones(n - 1), np.ones(n - 1), np.zeros(n), dtype=np.float32)
  state = tmps.FiniteMPS.random([2] * n, [16] * (n - 1), np.float32, hip, seed=21)
  dmrg = tmps.FiniteDMRG(state, mpo)
  e = dmrg.run_two_site(max_bond_dim=48, num_sweeps=4, num_krylov_vecs=12)
  np.testing.assert_allclose(e, eta[0], atol=2e-4 * abs(eta[0]))
  assert max(state.bond_dimensions) <= 48


@pytest.mark.parametrize("tag", cases.MPS_GOLDEN_TAGS)
def test_mps_measurements_match_reference_golden_on_device(hip, tag):
  """Reduced density matrices, transfer operators, <O>, <O1 O2>, one-site gates on device tensors vs the
  numbers recorded from the reference's FiniteMPS (tests/golden/make_golden_mps.py)."""
  cases.check_mps_golden_case(hip, cases.load_mps_golden(), tag, 1e-10)


@pytest.mark.parametrize("tag", cases.INFINITE_MPS_GOLDEN_TAGS)
def test_infinite_mps_matches_reference_golden_on_device(hip, tag):
  """InfiniteMPS.canonicalize end to end in HBM: Krylov-Schur `eigs` on device vectors for the dominant
  transfer-matrix eigenvectors, Hermitian `eigh`, int32 masks + `index_update` for the pseudo-inverse,
  truncated `svd`, `inv`; Schmidt spectrum vs the reference's (golden_mps.npz)."""
  cases.check_infinite_mps_golden_case(hip, cases.load_mps_golden(), tag, 1e-11)


def test_free_fermion_2d_one_site_dmrg_on_device(hip):
  """mpo_test.py:94-128 on the GPU: FiniteFreeFermion2D (2 x 4 grid, ancillary dimension 6) + one-site DMRG."""
  from tensornetwork_amd import mpo as tmpo
  n1, n2, D = 2, 4, 16
  model = tmpo.FiniteFreeFermion2D(-1.0, -1.0, -1.0, n1, n2, np.float64, backend=hip)
  assert all(isinstance(t, ta.DeviceTensor) for t in model)
  state = tmps.FiniteMPS.random([2] * (n1 * n2), [D] * (n1 * n2 - 1), np.float64, hip, seed=5)
  energy = tmps.FiniteDMRG(state, model).run_one_site(num_sweeps=6, precision=1e-10)
  n = n1 * n2
  tij = -np.eye(n)
  for s in range(n):
    col, row = divmod(s, n1)
    if row < n1 - 1:
      tij[s, s + 1] = tij[s + 1, s] = -1.0
    if col < n2 - 1:
      tij[s, s + n1] = tij[s + n1, s] = -1.0
  np.testing.assert_allclose(energy, min(np.cumsum(np.linalg.eigvalsh(tij))), rtol=1e-6)
