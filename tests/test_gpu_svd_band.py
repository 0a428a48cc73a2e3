"""K7b (tnh_svd_band_*): band reduction + spectrum slicing + inverse iteration, on the MI355X.

Oracle: np.linalg.svd in float64 of the same float32 input (what decompositions.py:36 calls), plus the
stage-by-stage NumPy statement of the algorithm in tools/svd_band_model.py.  Tolerances (f32 path):
|s - s_ref| <= 1e-5 s_0 for ALL values (s_rest included), orthonormality 1e-4, reconstruction within
1e-4 s_0 of the best rank-k approximation error."""
import ctypes
import os
import sys

import numpy as np
import pytest

import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def gaussian(m, n, seed):
  return np.random.default_rng(seed).standard_normal((m, n)).astype(np.float32)


def graded(m, n, seed, rate=32.0):
  """decompositions_test.py:55-66 construction: prescribed spectrum between Haar factors."""
  rng = np.random.default_rng(seed)
  r = min(m, n)
  qu, _ = np.linalg.qr(rng.standard_normal((m, r)))
  qv, _ = np.linalg.qr(rng.standard_normal((n, r)))
  return ((qu * 2.0 ** (-np.arange(r) / rate)) @ qv.T).astype(np.float32)


def check_svd(a, u, s, vh, s_rest, k, tag=""):
  a64 = a.astype(np.float64)
  ur, sr, vr = np.linalg.svd(a64, full_matrices=False)
  s0 = sr[0]
  s_all = np.concatenate([np.asarray(s, dtype=np.float64), np.asarray(s_rest, dtype=np.float64)])
  assert s_all.shape == sr.shape
  assert np.max(np.abs(s_all - sr)) <= 1e-5 * s0, (tag, np.max(np.abs(s_all - sr)) / s0)
  u64, v64 = np.asarray(u, dtype=np.float64), np.asarray(vh, dtype=np.float64)
  assert u64.shape == (a.shape[0], k) and v64.shape == (k, a.shape[1])
  assert np.max(np.abs(u64.T @ u64 - np.eye(k))) <= 1e-4, (tag, "U", np.max(np.abs(u64.T @ u64 - np.eye(k))))
  assert np.max(np.abs(v64 @ v64.T - np.eye(k))) <= 1e-4, (tag, "V", np.max(np.abs(v64 @ v64.T - np.eye(k))))
  rec = np.linalg.norm(a64 - (u64 * s_all[:k]) @ v64)
  best = np.sqrt(np.sum(sr[k:] ** 2))
  assert rec <= best + 1e-4 * s0 * np.sqrt(k), (tag, rec, best)
  # every kept triplet is a singular triplet of A:  |A v - s u| small
  resid = np.max(np.linalg.norm(a64 @ v64.T - u64 * s_all[:k], axis=0))
  assert resid <= 2e-5 * s0, (tag, resid / s0)


@pytest.mark.parametrize("kind,m,n,k", [("gauss", 1024, 1024, 64), ("graded", 1024, 1024, 64),
                                        ("gauss", 2048, 1024, 128), ("gauss", 1024, 2048, 100),
                                        ("graded", 1536, 1280, 62)])
def test_band_svd_matches_lapack(hip, kind, m, n, k):
  a = (gaussian if kind == "gauss" else graded)(m, n, seed=m + n + k)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_singular_values=k)
  assert hip.last_svd_path == "band", (hip.last_svd_path, hip.last_svd_band_status)
  check_svd(a, u, s, vh, s_rest, k, f"{kind} {m}x{n} k={k}")


def test_band_svd_truncation_error_rule(hip):
  """decompositions.py:38-57 with max_truncation_error on the band path: keep = min(max_sv, #values whose tail
  norm exceeds the error)."""
  a = graded(1024, 1024, seed=5, rate=8.0)
  sr = np.linalg.svd(a.astype(np.float64), compute_uv=False)
  err = 1e-3
  trunc = np.sqrt(np.cumsum(sr[::-1] ** 2))
  want = int(min(200, np.count_nonzero(trunc > err * sr[0])))
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_singular_values=200, max_truncation_error=err,
                             relative=True)
  assert hip.last_svd_path == "band"
  assert s.shape[0] == want and s_rest.shape[0] == 1024 - want
  check_svd(a, u, s, vh, s_rest, want, "trunc-error rule")


def test_band_svd_falls_back_loudly_where_it_cannot_be_accurate(hip):
  """Rank-deficient input (panel Gram singular) and exactly repeated kept values: the device reports it and the
  backend re-runs the Jacobi path -- the result is still right, and the path taken is recorded."""
  rng = np.random.default_rng(3)
  low = (rng.standard_normal((1024, 8)) @ rng.standard_normal((8, 1024))).astype(np.float32)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(low), 1, max_singular_values=16)
  assert hip.last_svd_path.startswith("jacobi (band path reported status") and hip.last_svd_band_status != 0
  sr = np.linalg.svd(low.astype(np.float64), compute_uv=False)
  np.testing.assert_allclose(np.concatenate([np.asarray(s), np.asarray(s_rest)]), sr, atol=1e-5 * sr[0])
  # identical singular values: the kept vectors are only defined as a subspace
  q, _ = np.linalg.qr(rng.standard_normal((1024, 1024)))
  spec = np.concatenate([np.full(32, 2.0), np.linspace(1.0, 0.1, 1024 - 32)])
  q2, _ = np.linalg.qr(rng.standard_normal((1024, 1024)))
  deg = ((q * spec) @ q2.T).astype(np.float32)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(deg), 1, max_singular_values=32)
  # round 3: runs of kept values that f32 cannot tell apart are orthonormalised inside the band path (cluster
  # Gram-Schmidt on the inverse-iteration vectors): no fall-back needed
  assert hip.last_svd_path == "band", hip.last_svd_band_status
  check_svd(deg, u, s, vh, s_rest, 32, "degenerate")
  # pairs (the spectrum of a complex matrix's real embedding looks like this), cut in the middle of nothing
  spec2 = np.repeat(np.linspace(3.0, 0.05, 512), 2)
  pairs = ((q * spec2) @ q2.T).astype(np.float32)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(pairs), 1, max_singular_values=64)
  assert hip.last_svd_path == "band", hip.last_svd_band_status
  check_svd(pairs, u, s, vh, s_rest, 64, "pairs")


def test_band_stage_outputs_match_the_numpy_model(hip):
  """Stage by stage against tools/svd_band_model.py on one 512 x 512 input: the band has the singular values of A
  (orthogonal stage 1), T = B^T B, brackets contain LAPACK's values."""
  import svd_band_model as model
  n, kcap = 512, 32
  a = gaussian(n, n, seed=11)
  lib = hip.lib
  nbytes = ctypes.c_size_t(0)
  _lib.check(lib.tnh_svd_band_work_bytes(_lib.F32, n, n, kcap, ctypes.byref(nbytes)))
  work = DeviceTensor.empty((nbytes.value // 8 + 1,), _lib.F64)
  s_all = DeviceTensor.empty((n,), _lib.F32)
  da = hip.convert_to_tensor(a)
  status = ctypes.c_int(-1)
  _lib.check(lib.tnh_svd_band_factor(_lib.F32, n, n, ctypes.c_void_p(da.ptr), ctypes.c_void_p(s_all.ptr),
                                     ctypes.c_void_p(work.ptr), kcap, ctypes.byref(status)), "tnh_svd_band_factor")
  assert status.value == 0
  offs = (ctypes.c_int64 * 12)()
  _lib.check(lib.tnh_svd_band_layout(_lib.F32, n, n, kcap, offs, 12))
  names = ["Af", "Vl", "Vr", "Tl", "Tr", "Dblk", "Eblk", "Bd", "Tb", "lo", "hi", "X"]
  off = dict(zip(names, [int(x) for x in offs]))
  base = (work.ptr + 255) & ~255

  def fetch(name, count, dtype):
    out = np.empty(count, dtype=dtype)
    _lib.check(lib.tnh_d2h(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(base + off[name]), out.nbytes))
    return out

  bd = fetch("Bd", n * 17, np.float64).reshape(n, 17)
  tb = fetch("Tb", n * 17, np.float64).reshape(n, 17)
  sr = np.linalg.svd(a.astype(np.float64), compute_uv=False)
  sb = np.linalg.svd(model.band_dense(bd), compute_uv=False)
  assert np.max(np.abs(sb - sr)) <= 3e-6 * sr[0]          # stage 1 is an orthogonal equivalence (f32 rounding)
  np.testing.assert_allclose(tb, model.gram_band(bd), rtol=1e-13, atol=1e-13 * sr[0] ** 2)
  lo = fetch("lo", n, np.float64)
  hi = fetch("hi", n, np.float64)
  sb_asc = sb[::-1]
  assert np.all(lo <= sb_asc + 1e-9 * sr[0]) and np.all(sb_asc <= hi + 1e-9 * sr[0])
  assert np.max(hi - lo) <= 4e-6 * sr[0]        # 20 bits of the Gershgorin bound on sigma_max
  np.testing.assert_allclose(np.asarray(s_all), sr, atol=5e-6 * sr[0])
  # the model's own stage 1 (same formulas on the host) gives a band with the same singular values
  af, _, _, ok = model.to_band(a)
  assert ok
  sm = np.linalg.svd(model.band_dense(model.band_of(af)), compute_uv=False)
  assert np.max(np.abs(sm - sb)) <= 3e-6 * sr[0]


def test_band_dpp_broadcasts_equal_shuffles(hip, monkeypatch):
  """The 16-lane LDL^T / solve kernels move pivots with DPP row_newbcast / row_ror; TNH_SVDB_DPP=0 builds the same
  kernels on __shfl.  Counts are integers, so the singular values (bracket midpoints) are bit-identical; the vectors
  agree up to the summation order of the 16-lane dot products."""
  a = gaussian(1024, 1024, seed=21)
  d = hip.convert_to_tensor(a)
  monkeypatch.setenv("TNH_SVDB_DPP", "1")
  u1, s1, v1, r1 = hip.svd(d, 1, max_singular_values=32)
  assert hip.last_svd_path == "band"
  monkeypatch.setenv("TNH_SVDB_DPP", "0")
  u2, s2, v2, r2 = hip.svd(d, 1, max_singular_values=32)
  assert hip.last_svd_path == "band"
  np.testing.assert_array_equal(np.asarray(s1), np.asarray(s2))
  np.testing.assert_array_equal(np.asarray(r1), np.asarray(r2))
  np.testing.assert_allclose(np.asarray(u1), np.asarray(u2), atol=2e-6)
  np.testing.assert_allclose(np.asarray(v1), np.asarray(v2), atol=2e-6)


def test_split_node_4096_config(hip):
  """BASELINE configs[2] through the Node API: (16,)*6 node split 3|3, keep 256; prescribed spectrum
  s_i = 2^(-i/32) (decompositions_test.py:55-66) in the mixed edge order of split_node_test.py:36-47."""
  rng = np.random.default_rng(0)
  n = 4096
  qu, _ = np.linalg.qr(rng.standard_normal((n, n)))
  qv, _ = np.linalg.qr(rng.standard_normal((n, n)))
  mat = ((qu * 2.0 ** (-np.arange(n) / 32.0)) @ qv.T).astype(np.float32)
  # node axes (a0 a1 a2 b0 b1 b2); hand split_node the mixed order left = [a2, a0, a1], right = [b1, b2, b0]
  node = ta.Node(hip.convert_to_tensor(mat.reshape((16,) * 6)), backend=hip)
  left = [node[2], node[0], node[1]]
  right = [node[4], node[5], node[3]]
  l, r, _ = ta.split_node(node, left, right, max_singular_values=256)
  assert hip.last_svd_path == "band", hip.last_svd_band_status
  full = np.tensordot(np.asarray(l.tensor, dtype=np.float64), np.asarray(r.tensor, dtype=np.float64), [[3], [0]])
  # l axes: a2 a0 a1 | bond ; r axes: bond | b1 b2 b0  ->  back to (a0 a1 a2 b0 b1 b2)
  full = full.transpose(1, 2, 0, 5, 3, 4).reshape(n, n)
  sr = np.linalg.svd(mat.astype(np.float64), compute_uv=False)
  best = np.sqrt(np.sum(sr[256:] ** 2))
  assert np.linalg.norm(full - mat) <= best + 2e-3 * sr[0]


def test_complex64_svd_through_the_real_embedding(hip):
  """complex64 split through the band path: the real embedding doubles every singular value, the
  band path returns orthonormal bases of the doubled subspaces, k complex directions are extracted on the way out
  (split_node_test.py:63-73 is the small-matrix version of this call)."""
  rng = np.random.default_rng(12)
  m, n, k = 1024, 768, 48
  a = (rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))).astype(np.complex64)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_singular_values=k)
  assert "complex via the real embedding" in hip.last_svd_path, (hip.last_svd_path, hip.last_svd_band_status)
  assert u.dtype == np.complex64 and s.dtype == np.complex64 and s_rest.shape == (n - k,)
  a128 = a.astype(np.complex128)
  sr = np.linalg.svd(a128, compute_uv=False)
  s_all = np.concatenate([np.asarray(s).real, np.asarray(s_rest).real]).astype(np.float64)
  assert np.max(np.abs(np.asarray(s).imag)) == 0.0
  assert np.max(np.abs(s_all - sr)) <= 1e-5 * sr[0]
  uu, vv = np.asarray(u).astype(np.complex128), np.asarray(vh).astype(np.complex128)
  assert np.max(np.abs(uu.conj().T @ uu - np.eye(k))) <= 1e-4
  assert np.max(np.abs(vv @ vv.conj().T - np.eye(k))) <= 1e-4
  assert np.max(np.linalg.norm(a128 @ vv.conj().T - uu * s_all[:k], axis=0)) <= 2e-5 * sr[0]
  rec = np.linalg.norm(a128 - (uu * s_all[:k]) @ vv)
  assert rec <= np.sqrt(np.sum(sr[k:] ** 2)) + 1e-4 * sr[0] * np.sqrt(k)


# ---- round 4: every call shape the reference makes takes the band path ------------------------------
def test_band_svd_truncation_error_alone(hip):
  """split_node(max_truncation_err=...) without max_singular_values (network_operations.py:130-137, 219-223): the
  values come first, k is picked on the host by decompositions.py:38-57, then the vectors."""
  a = graded(1024, 768, seed=21, rate=16.0)
  sr = np.linalg.svd(a.astype(np.float64), compute_uv=False)
  for err, relative in ((2e-3, False), (1e-2, True)):
    trunc = np.sqrt(np.cumsum(sr[::-1] ** 2))
    want = int(np.count_nonzero(trunc > (err * sr[0] if relative else err)))
    u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_truncation_error=err, relative=relative)
    assert hip.last_svd_path == "band", (hip.last_svd_path, hip.last_svd_band_status)
    assert abs(s.shape[0] - want) <= 1 and s_rest.shape[0] == 768 - s.shape[0]     # a value ON the threshold may fall either way
    check_svd(a, u, s, vh, s_rest, s.shape[0], f"trunc error alone {err} rel={relative}")


@pytest.mark.parametrize("m,n", [(1024, 1024), (1280, 768)])
def test_band_svd_full(hip, m, n):
  """split_node_full_svd (network_operations.py:446-588): no truncation at all -- k = min(m, n) vectors."""
  a = gaussian(m, n, seed=m + 3 * n)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1)
  assert hip.last_svd_path == "band", (hip.last_svd_path, hip.last_svd_band_status)
  assert s_rest.shape == (0,)
  check_svd(a, u, s, vh, s_rest, min(m, n), f"full {m}x{n}")
  # the whole thing reconstructs A (not only the best rank-k part)
  rec = (np.asarray(u, dtype=np.float64) * np.asarray(s, dtype=np.float64)) @ np.asarray(vh, dtype=np.float64)
  assert np.max(np.abs(rec - a)) <= 2e-5 * float(np.asarray(s)[0])


def test_band_svd_keeps_more_than_half(hip):
  a = gaussian(1024, 1024, seed=77)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_singular_values=700)
  assert hip.last_svd_path == "band", (hip.last_svd_path, hip.last_svd_band_status)
  check_svd(a, u, s, vh, s_rest, 700, "k = 700 of 1024")


@pytest.mark.parametrize("kind,m,n,k", [("gauss", 1000, 1000, 50), ("graded", 1030, 900, 64), ("gauss", 777, 1500, 33)])
def test_band_svd_sides_that_are_not_multiples_of_16(hip, kind, m, n, k):
  """min(m, n) % 16 != 0: blockdiag(A, delta I) padding (HipBackend._svd_band_pad); D = 500 bonds give 1000 x 1000."""
  a = (gaussian if kind == "gauss" else graded)(m, n, seed=m + n + k)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_singular_values=k)
  assert hip.last_svd_path == "band", (hip.last_svd_path, hip.last_svd_band_status)
  assert s_rest.shape == (min(m, n) - k,)
  check_svd(a, u, s, vh, s_rest, k, f"padded {kind} {m}x{n} k={k}")
  # and with the truncation rule on the padded spectrum
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_singular_values=k, max_truncation_error=1e-3, relative=True)
  assert hip.last_svd_path == "band"
  sr = np.linalg.svd(a.astype(np.float64), compute_uv=False)
  want = int(min(k, np.count_nonzero(np.sqrt(np.cumsum(sr[::-1] ** 2)) > 1e-3 * sr[0])))
  assert abs(s.shape[0] - want) <= 1
  check_svd(a, u, s, vh, s_rest, s.shape[0], "padded + truncation rule")


def test_band_svd_kept_values_keep_their_relative_accuracy(hip):
  """the kept values come from the brackets the vectors stage refines to 2^-32 s_1, not from the
  20-bit brackets every value gets -- a kept value of 2e-5 s_1 is good to ~1e-3 relative (the f32 reduction's own
  backward error, 1e-7 s_1 absolute), not to the 3 % a 5e-7 s_1 bracket would give."""
  rng = np.random.default_rng(9)
  n, k = 1024, 64
  q1, _ = np.linalg.qr(rng.standard_normal((n, n)))
  q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
  spec = np.concatenate([2.0 ** (-np.arange(k) / 4.0), np.full(n - k, 2.0 ** -18)])      # kept: 1 ... 1.8e-5
  a = ((q1 * spec) @ q2.T).astype(np.float32)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_singular_values=k)
  assert hip.last_svd_path == "band", (hip.last_svd_path, hip.last_svd_band_status)
  sr = np.linalg.svd(a.astype(np.float64), compute_uv=False)
  got = np.asarray(s, dtype=np.float64)
  assert np.max(np.abs(got - sr[:k])) <= 2e-7 * sr[0], np.max(np.abs(got - sr[:k])) / sr[0]
  assert np.max(np.abs(got - sr[:k]) / sr[:k]) <= 1e-2


# ---- round 4: float64 (and complex128 through it) on the band path ------------------------------------------------------
def check_svd64(a, u, s, vh, s_rest, k, tag="", tol_s=1e-11, tol_rest=3e-8, tol_orth=1e-9, tol_res=2e-10):
  """f64 band path vs LAPACK.  Stated tolerances: kept values 1e-11 s_0 (Rayleigh quotients on
  the band), DISCARDED values 3e-8 s_0 -- brackets of 28 bits (2e-9 s_0) where the un-pivoted LDL^T of T = B^T B
  resolves them, and a floor of sqrt(eps64) s_0 ~ 1.5e-8 s_0 below which T = B^T B cannot tell a value from zero
  (measured: 1.5e-9 on Gaussian input, 1.6e-8 on the tail of a graded one) --, orthonormality 1e-9, triplet residual
  |A v - s u| <= 2e-10 s_0."""
  cplx = np.iscomplexobj(a)
  a128 = a.astype(np.complex128 if cplx else np.float64)
  sr = np.linalg.svd(a128, compute_uv=False)
  s0 = sr[0]
  sk, srest = np.real(np.asarray(s)).astype(np.float64), np.real(np.asarray(s_rest)).astype(np.float64)
  assert sk.shape == (k,) and srest.shape == (min(a.shape) - k,)
  assert np.max(np.abs(sk - sr[:k])) <= tol_s * s0, (tag, "kept", np.max(np.abs(sk - sr[:k])) / s0)
  if srest.size:
    assert np.max(np.abs(srest - sr[k:])) <= tol_rest * s0, (tag, "rest", np.max(np.abs(srest - sr[k:])) / s0)
  uu, vv = np.asarray(u).astype(a128.dtype), np.asarray(vh).astype(a128.dtype)
  assert uu.shape == (a.shape[0], k) and vv.shape == (k, a.shape[1])
  assert np.max(np.abs(uu.conj().T @ uu - np.eye(k))) <= tol_orth, (tag, "U", np.max(np.abs(uu.conj().T @ uu - np.eye(k))))
  assert np.max(np.abs(vv @ vv.conj().T - np.eye(k))) <= tol_orth, (tag, "V", np.max(np.abs(vv @ vv.conj().T - np.eye(k))))
  resid = np.max(np.linalg.norm(a128 @ vv.conj().T - uu * sk, axis=0))
  assert resid <= tol_res * s0, (tag, "resid", resid / s0)


def graded64(m, n, seed, rate=32.0):
  rng = np.random.default_rng(seed)
  r = min(m, n)
  qu, _ = np.linalg.qr(rng.standard_normal((m, r)))
  qv, _ = np.linalg.qr(rng.standard_normal((n, r)))
  return (qu * 2.0 ** (-np.arange(r) / rate)) @ qv.T


@pytest.mark.parametrize("kind,m,n,k", [("gauss", 1024, 1024, 64), ("graded", 1024, 1024, 64), ("gauss", 1536, 1024, 100),
                                        ("gauss", 1024, 2048, 128), ("graded", 1000, 1100, 50)])
def test_band_svd_float64(hip, kind, m, n, k):
  """The reference's default dtype (ncon_interface_test.py:39-43, base_mps.py) on the band path: f64 panels by two
  Cholesky-QR passes, values to 28 bits, kept values as Rayleigh quotients, one Newton-Schulz step on the kept
  vectors."""
  a = np.random.default_rng(m + n + k).standard_normal((m, n)) if kind == "gauss" else graded64(m, n, seed=m + n + k)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_singular_values=k)
  assert hip.last_svd_path == "band", (hip.last_svd_path, hip.last_svd_band_status)
  assert u.dtype == np.float64 and s.dtype == np.float64 and vh.dtype == np.float64 and s_rest.dtype == np.float64
  check_svd64(a, u, s, vh, s_rest, k, f"f64 {kind} {m}x{n} k={k}", tol_rest=3e-8 if min(m, n) % 16 == 0 else 8e-8)


def test_band_svd_float64_other_call_shapes(hip):
  a = graded64(1024, 1024, seed=4, rate=16.0)
  sr = np.linalg.svd(a, compute_uv=False)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_truncation_error=1e-3, relative=True)
  assert hip.last_svd_path == "band", (hip.last_svd_path, hip.last_svd_band_status)
  want = int(np.count_nonzero(np.sqrt(np.cumsum(sr[::-1] ** 2)) > 1e-3 * sr[0]))
  assert abs(s.shape[0] - want) <= 1
  check_svd64(a, u, s, vh, s_rest, s.shape[0], "f64 trunc error alone")
  g = np.random.default_rng(8).standard_normal((768, 768))
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(g), 1)                     # split_node_full_svd
  assert hip.last_svd_path == "band", (hip.last_svd_path, hip.last_svd_band_status)
  check_svd64(g, u, s, vh, s_rest, 768, "f64 full", tol_res=5e-9)
  rec = (np.asarray(u) * np.asarray(s)) @ np.asarray(vh)
  assert np.max(np.abs(rec - g)) <= 1e-9 * sr[0] * 100
  # a kept value below 1e-5 s_1 is outside the f64 band path's range: reported, Jacobi answers
  steep = graded64(1024, 1024, seed=5, rate=4.0)
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(steep), 1, max_singular_values=128)      # s_128 = 2^-32 s_1
  assert hip.last_svd_path.startswith("jacobi") and hip.last_svd_band_status & 16
  np.testing.assert_allclose(np.asarray(s), np.linalg.svd(steep, compute_uv=False)[:128], rtol=1e-7, atol=1e-14)


def test_band_svd_complex128_through_the_f64_embedding(hip):
  rng = np.random.default_rng(13)
  m, n, k = 768, 640, 40
  a = rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
  u, s, vh, s_rest = hip.svd(hip.convert_to_tensor(a), 1, max_singular_values=k)
  assert "complex via the real embedding" in hip.last_svd_path, (hip.last_svd_path, hip.last_svd_band_status)
  assert u.dtype == np.complex128 and s.dtype == np.complex128
  check_svd64(a, u, s, vh, s_rest, k, "complex128")


def test_qr_fast_path_is_not_taken_inside_a_graph_capture(hip):
  """the 16-wide-panel QR reads a status word back (a stream synchronisation), illegal while the stream is
  captured -- a captured QR takes the column path and replays correctly."""
  a = gaussian(512, 256, seed=3)
  q0, r0 = hip.qr(hip.convert_to_tensor(a), 1)
  ref_q, ref_r = np.asarray(q0), np.asarray(r0)
  x = hip.convert_to_tensor(a)
  graph = hip.capture(lambda t: hip.qr(t, 1), x)
  q, r = graph.launch()
  hip.synchronize()
  np.testing.assert_allclose(np.asarray(q) @ np.asarray(r), a, atol=2e-5)
  np.testing.assert_allclose(np.abs(np.asarray(r)), np.abs(ref_r), atol=2e-4)
  assert np.max(np.abs(np.asarray(q).T @ np.asarray(q) - np.eye(256))) < 1e-4
  del ref_q


def test_band_svd_backoff_is_an_inspectable_resettable_per_backend_policy(hip):
  """which path a call takes after the band path has reported on a shape is a documented policy of
  the backend OBJECT: visible (`svd_band_policy`), resettable, switchable, and named in `last_svd_path` -- never a
  silent function of the process's history.  Both paths obey the same truncation rule."""
  rng = np.random.default_rng(8)
  low = (rng.standard_normal((1024, 8)) @ rng.standard_normal((8, 1024))).astype(np.float32)
  sr = np.linalg.svd(low.astype(np.float64), compute_uv=False)
  dev = hip.convert_to_tensor(low)
  hip.reset_svd_band_policy()
  assert hip.svd_band_policy()["backoff"] == {} and hip.svd_band_policy()["status_read_early_for"] == []
  paths = []
  for _ in range(4):
    _, s, _, _ = hip.svd(dev, 1, max_singular_values=16)
    paths.append(hip.last_svd_path)
    np.testing.assert_allclose(np.asarray(s)[:8], sr[:8], rtol=1e-4)
  # first two calls try the band path and report; from the second consecutive report on the shape is skipped
  assert paths[0].startswith("jacobi (band path reported status") and paths[1].startswith("jacobi (band path reported status")
  assert paths[2].startswith("jacobi (band path skipped") and paths[3].startswith("jacobi (band path skipped")
  state = hip.svd_band_policy()
  assert len(state["backoff"]) == 1 and list(state["backoff"].values())[0]["consecutive_reports"] == 2
  # reset: the next call tries the band path again; with the policy switched off it always tries
  hip.reset_svd_band_policy()
  hip.svd(dev, 1, max_singular_values=16)
  assert hip.last_svd_path.startswith("jacobi (band path reported status")
  hip.svd_band_backoff = False
  try:
    for _ in range(3):
      hip.svd(dev, 1, max_singular_values=16)
      assert hip.last_svd_path.startswith("jacobi (band path reported status")
  finally:
    hip.svd_band_backoff = True
    hip.reset_svd_band_policy()
  # a well-conditioned matrix of the same shape is untouched by any of this
  good = rng.standard_normal((1024, 1024)).astype(np.float32)
  hip.svd(hip.convert_to_tensor(good), 1, max_singular_values=16)
  assert hip.last_svd_path == "band"


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_complex_svd_cluster_branch_on_gpu(hip, dtype):
  """the branch of the complex band path that handles degenerate complex values (greedy
  Gram-Schmidt over all 2k real-embedding candidates on their Gram matrix) had never run on the GPU.  Forced here
  on an ordinary matrix -- it must give the same decomposition as the Newton-Schulz branch and as LAPACK."""
  rng = np.random.default_rng(21)
  m, n, k = 768, 640, 24
  a = (rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))).astype(dtype)
  dev = hip.convert_to_tensor(a)
  u0, s0, vh0, _ = hip.svd(dev, 1, max_singular_values=k)
  assert "complex via the real embedding" in hip.last_svd_path, (hip.last_svd_path, hip.last_svd_band_status)
  hip.svd_complex_force_cluster_path = True
  try:
    u, s, vh, s_rest = hip.svd(dev, 1, max_singular_values=k)
    assert "complex via the real embedding" in hip.last_svd_path
  finally:
    hip.svd_complex_force_cluster_path = False
  a128 = a.astype(np.complex128)
  sr = np.linalg.svd(a128, compute_uv=False)
  tol = 1e-5 if dtype == np.complex64 else 1e-12
  s_all = np.concatenate([np.asarray(s).real, np.asarray(s_rest).real]).astype(np.float64)
  assert np.max(np.abs(s_all[:k] - sr[:k])) <= tol * sr[0]
  np.testing.assert_array_equal(np.asarray(s), np.asarray(s0))
  uu, vv = np.asarray(u).astype(np.complex128), np.asarray(vh).astype(np.complex128)
  assert np.max(np.abs(uu.conj().T @ uu - np.eye(k))) <= 20 * tol
  assert np.max(np.abs(vv @ vv.conj().T - np.eye(k))) <= 20 * tol
  assert np.max(np.linalg.norm(a128 @ vv.conj().T - uu * s_all[:k], axis=0)) <= 20 * tol * sr[0]
  # same subspaces as the Newton-Schulz branch (the vectors themselves differ by phases)
  p0 = np.asarray(u0).astype(np.complex128)
  assert np.linalg.norm(uu @ (uu.conj().T @ p0) - p0) <= 100 * tol * np.sqrt(k)


def test_fast_band_reduction_is_taken_guarded_and_backs_off(monkeypatch):
  """Round 6 (tnh_svd_band_fast.inc): which band reduction a call ran is observable (tnh_svd_band_last_stage1) --
  1 the four-launch fast stage (f32 and f64), 0 the loop of rounds 3-5 (TNH_SVDB_FAST=0), 2 the fast stage reported an
  ill-conditioned panel (a numerically half-rank matrix) and the stage was repeated with that loop, 3 the loop
  directly while the shape's back-off lasts -- and the results obey the same tolerances whichever ran."""
  be = ta.get_hip_backend()
  last = be.lib.tnh_svd_band_last_stage1
  rng = np.random.default_rng(77)
  a = gaussian(1040, 1024, 5)                    # (a shape no other test uses: the back-off table is per shape)
  u, s, vh, rest = be.svd(be.convert_to_tensor(a), 1, max_singular_values=64)
  assert be.last_svd_path == "band" and last() == 1
  check_svd(a, u, s, vh, rest, 64, "fast f32")
  u, s, vh, rest = be.svd(be.convert_to_tensor(a.astype(np.float64)), 1, max_singular_values=64)
  assert be.last_svd_path == "band" and last() == 1
  sr = np.linalg.svd(a.astype(np.float64), compute_uv=False)
  assert np.max(np.abs(np.asarray(s) - sr[:64])) <= 1e-12 * sr[0]
  monkeypatch.setenv("TNH_SVDB_FAST", "0")
  u0, s0, vh0, rest0 = be.svd(be.convert_to_tensor(a), 1, max_singular_values=64)
  assert last() == 0
  check_svd(a, u0, s0, vh0, rest0, 64, "accurate loop")
  monkeypatch.delenv("TNH_SVDB_FAST")
  # half of the spectrum is zero: panels of the transition are ill-conditioned
  m = 1056
  qu, _ = np.linalg.qr(rng.standard_normal((m, m)))
  qv, _ = np.linalg.qr(rng.standard_normal((m, m)))
  spec = np.where(np.arange(m) < m // 2, np.linspace(2, 1, m), 0.0)
  h = ((qu * spec) @ qv.T).astype(np.float32)
  seen = []
  for _ in range(3):
    u, s, vh, rest = be.svd(be.convert_to_tensor(h), 1, max_singular_values=64)
    assert be.last_svd_path == "band"
    seen.append(last())
    check_svd(h, u, s, vh, rest, 64, "half rank")
  assert seen[0] == 2 and seen[1] == 3 and seen[2] == 3, seen
