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


def test_three_way_bf16_split_arithmetic():
  """The arithmetic behind the f32-on-bf16-cores GEMM (csrc/tnh_gemm.hip, f32_split3_*): hi = bf16(x),
  mid = bf16(x - hi), lo = bf16(x - hi - mid) reproduce a normal float32 exactly, and the six products kept by
  the kernel approximate a * b to ~2^-25 |a||b| -- restated with the oracle's bf16 rounding."""
  from oracle import numpy_oracle as orc
  rng = np.random.default_rng(5)
  x = (rng.standard_normal(200000) * np.exp(rng.uniform(-60, 60, 200000))).astype(np.float32)
  x = np.concatenate([x, np.float32([0.0, -0.0, 1.0, -1.0, 3.0e38, 1.1754944e-38, 1 + 2**-23, 1 - 2**-24])])

  def split(v):
    hi = orc.round_bf16(v).astype(np.float32)
    r1 = (v - hi).astype(np.float32)
    mid = orc.round_bf16(r1).astype(np.float32)
    r2 = (r1 - mid).astype(np.float32)
    lo = orc.round_bf16(r2).astype(np.float32)
    return hi, mid, lo

  hi, mid, lo = split(x)
  # the two subtractions are exact in float32, so this is an identity for every normal input
  assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
  assert np.all(np.abs(mid) <= np.abs(hi) * 2.0**-8 + 1e-45) and np.all(np.abs(lo) <= np.abs(hi) * 2.0**-16 + 1e-45)
  y = rng.standard_normal(x.size).astype(np.float32)
  yh, ym, yl = split(y)
  f = lambda p, q: p.astype(np.float64) * q.astype(np.float64)
  six = f(lo, yh) + f(hi, yl) + f(mid, ym) + f(mid, yh) + f(hi, ym) + f(hi, yh)
  exact = f(x, y)
  ok = np.abs(exact) > 0
  assert np.max(np.abs(six - exact)[ok] / np.abs(exact)[ok]) < 2.0**-24     # dropped terms: mid*lo, lo*mid, lo*lo
  three = f(mid, yh) + f(hi, ym) + f(hi, yh)
  assert np.max(np.abs(three - exact)[ok] / np.abs(exact)[ok]) > 2.0**-18    # why six products and not three


@pytest.mark.parametrize("nb", [8, 16, 24, 128])
@pytest.mark.parametrize("groups", [1, 2, 4])
def test_svd_block_schedule_covers_every_pair_once(nb, groups):
  """The sweep schedules of the block Jacobi (circle method; grouped schedule on 2 / 4 streams,
  tnh_svd_block.hip): nb - 1 rounds, every block exactly once per round, every block pair exactly once per
  sweep, and the groups of a round never share a block (they run unsynchronised on different streams)."""
  import ctypes
  from tensornetwork_amd import _lib
  lib = _lib.load_library()
  if groups > 1 and (nb % (2 * groups) != 0 or nb < 4 * groups):
    out = np.zeros(((nb - 1), nb // 2, 2), dtype=np.int32)
    rounds = ctypes.c_int(0)
    assert lib.tnh_svd_block_schedule(nb, groups, out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(rounds)) != 0
    return
  out = np.full(((nb - 1), nb // 2, 2), -1, dtype=np.int32)
  rounds = ctypes.c_int(0)
  _lib.check(lib.tnh_svd_block_schedule(nb, groups, out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(rounds)),
             "tnh_svd_block_schedule")
  assert rounds.value == nb - 1
  seen = set()
  per_group = (nb // 2) // groups
  for r in range(nb - 1):
    blocks = out[r].reshape(-1)
    assert sorted(blocks.tolist()) == list(range(nb)), (r, blocks)
    for g in range(groups):
      mine = set(out[r, g * per_group:(g + 1) * per_group].reshape(-1).tolist())
      rest = set(np.delete(out[r], np.s_[g * per_group:(g + 1) * per_group], axis=0).reshape(-1).tolist())
      assert not (mine & rest)
    for a, b in out[r]:
      key = (min(a, b), max(a, b))
      assert a != b and key not in seen, (r, key)
      seen.add(key)
  assert len(seen) == nb * (nb - 1) // 2
