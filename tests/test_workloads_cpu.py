"""BASELINE workload builders (tensornetwork_amd.workloads) on the CPU oracle backend:
the reference's own known answers for these networks."""
import numpy as np
import pytest

from tensornetwork_amd import contractors, distributed, workloads as wl
from oracle.numpy_oracle import OracleBackend


@pytest.fixture(scope="module", params=["oracle", "hip-emulated"])
def be(request):
  """Both backends: the oracle, and HipBackend on the emulated C ABI (tests/emu_tnh.py: the product's backend code)."""
  if request.param == "oracle":
    yield OracleBackend()
    return
  from emu_tnh import emulated_backend  # pylint: disable=import-outside-toplevel
  with emulated_backend() as hip:
    yield hip


def test_wavelet_mera_energy_kat(be):
  """simple_mera_test.py:48-56: 20 descents of the maximally mixed state through the D=2
  wavelet MERA, then tr(rho h) and the layer energy network both give -1.242."""
  h = wl.ham_ising()
  w, u = wl.wavelet_mera_tensors()
  s = (np.eye(8) / 8).reshape((2,) * 6)
  for _ in range(20):
    s = np.asarray(wl.mera_descend(be, s, w, u,
                                   lambda nodes, order: contractors.greedy(nodes, output_edge_order=order)))
  en = np.trace(s.reshape(8, 8) @ h.reshape(8, 8))
  assert np.isclose(en, -1.242, rtol=1e-3, atol=1e-3)
  en2 = float(np.asarray(wl.mera_energy(be, h, s, w, u, lambda nodes: contractors.branch(nodes, nbranch=2))))
  assert np.isclose(en2, -1.242, rtol=1e-3, atol=1e-3)
  # descended state stays a unit-trace symmetric matrix
  m = s.reshape(8, 8)
  assert np.isclose(np.trace(m), 1.0) and np.allclose(m, m.T, atol=1e-12)


def test_mera_isometry_constraints(be):
  ham, rho, iso, dis = wl.mera_random_tensors(3, dtype=np.float64)
  w = iso.reshape(9, 3)
  np.testing.assert_allclose(w.T @ w, np.eye(3), atol=1e-12)
  u = dis.reshape(9, 9)
  np.testing.assert_allclose(u.T @ u, np.eye(9), atol=1e-12)
  # energy of the identity hamiltonian through isometric layers is tr(rho) = 1
  eye = np.eye(27).reshape((3,) * 6)
  en = float(np.asarray(wl.mera_energy(be, eye, rho, iso, dis, contractors.greedy)))
  assert np.isclose(en, 1.0, atol=1e-10)


def test_mps_overlap_is_norm(be):
  kets = wl.mps_tensors(6, 2, 4, dtype=np.float64)
  val = float(np.asarray(contractors.greedy(wl.mps_overlap_network(be, kets)).tensor))
  psi = kets[0]
  for k in kets[1:]:
    psi = np.tensordot(psi, k, [[psi.ndim - 1], [0]])
  assert np.isclose(val, np.sum(psi * psi), rtol=1e-12)


def test_regular_graph_is_3_regular():
  edges = wl.regular_graph_edges(64, 3, 6)
  assert len(edges) == 96 and len(set(edges)) == 96
  deg = np.zeros(64, int)
  for a, b in edges:
    assert a != b
    deg[a] += 1
    deg[b] += 1
  assert np.all(deg == 3)


def test_regular_network_sliced_matches_greedy(be):
  nodes = wl.random_regular_network(be, n=12, D=3, dtype=np.float64)
  ref = float(np.asarray(contractors.greedy(nodes).tensor))
  nodes = wl.random_regular_network(be, n=12, D=3, dtype=np.float64)
  cuts = distributed.choose_cut_edges(nodes, min_slices=9)
  out = float(np.asarray(distributed.contract_sliced(nodes, cuts)))
  assert np.isclose(out, ref, rtol=1e-10)
