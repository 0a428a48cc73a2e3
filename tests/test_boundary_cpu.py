"""The drop-in boundary, checkable WITHOUT the reference on the box: HipBackend against the
committed snapshot of the reference's interface (tests/golden/abstract_backend_signatures.json, written by
tests/golden/make_golden_signatures.py from abstract_backend.py:27-1046 and numpy_backend.py), and tensors of more
than 16 axes (one K1 / K5 launch indexes TNH_MAX_RANK = 16: coalescing pre-pass, several passes when needed)."""
import numpy as np
import pytest

from emu_tnh import emulated_backend
import cases as C
from tensornetwork_amd import hip_backend as hb


def test_hip_backend_accepts_the_reference_signatures_on_the_emulated_abi():
  with emulated_backend() as be:
    assert C.signature_mismatches(type(be)) == []


def test_permutation_plans_cover_any_rank():
  rng = np.random.default_rng(0)

  def run(x, passes, out_shape):
    cur = x.reshape(-1)
    for cshape, cperm in passes:
      assert len(cshape) <= hb.MAX_KERNEL_RANK
      cur = np.transpose(cur.reshape(cshape), cperm).copy().reshape(-1)
    return cur.reshape(out_shape)

  worst = 0
  for _ in range(200):
    nd = int(rng.integers(1, 23))
    shape = [int(rng.integers(1, 4)) if nd < 14 else int(rng.integers(1, 3)) for _ in range(nd)]
    perm = [int(p) for p in rng.permutation(nd)]
    x = np.arange(int(np.prod(shape))).reshape(shape)
    passes = hb._permute_passes(shape, perm)        # pylint: disable=protected-access
    worst = max(worst, len(passes))
    np.testing.assert_array_equal(run(x, passes, [shape[p] for p in perm]), np.transpose(x, perm))
  # rank <= 16 is always ONE pass; a full reversal of 22 two-level axes cannot be coalesced and takes two
  assert len(hb._permute_passes([2] * 16, list(range(15, -1, -1)))) == 1      # pylint: disable=protected-access
  shape, perm = [2] * 22, list(range(21, -1, -1))
  passes = hb._permute_passes(shape, perm)          # pylint: disable=protected-access
  assert len(passes) == 2
  x = np.arange(2 ** 22).reshape(shape)
  np.testing.assert_array_equal(run(x, passes, shape), np.transpose(x, perm))
  # axes that travel together are one axis: 20 axes, two blocks swapped -> a rank-2 transpose
  assert hb._coalesce_permutation([2] * 20, list(range(10, 20)) + list(range(10))) == ([1024, 1024], [1, 0])   # pylint: disable=protected-access


def test_high_rank_tensors_on_the_emulated_abi():
  with emulated_backend() as be:
    C.run_high_rank_cases(be)


def test_eigsh_defaults_follow_the_interface():
  """abstract_backend.py:380-391: numeig = 1, which = 'LR' (largest real part = scipy's 'LA' for a Hermitian operator)."""
  with emulated_backend() as be:
    rng = np.random.default_rng(3)
    h = rng.standard_normal((40, 40))
    h = (h + h.T).astype(np.float64)
    init = be.convert_to_tensor(rng.standard_normal(40))
    eta, vecs = be.eigsh(lambda x, m: be.tensordot(m, x, 1), [be.convert_to_tensor(h)], init, num_krylov_vecs=30, tol=1e-10)
    assert len(eta) == 1 and len(vecs) == 1
    np.testing.assert_allclose(float(np.asarray(eta[0])), np.linalg.eigvalsh(h)[-1], rtol=1e-8)
    with pytest.raises(ValueError):
      be.eigsh(lambda x: x, initial_state=init, which="SI")


def test_index_update_with_a_tensor_assignee_on_the_emulated_abi():
  with emulated_backend() as be:
    C.run_index_update_tensor_cases(be)
