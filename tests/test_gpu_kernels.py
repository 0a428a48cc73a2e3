"""GPU parity tests, kernel level: every C-ABI kernel family against the CPU oracle
on seeded inputs (bit-exact for layout work, stated tolerances for arithmetic).
All calls go through HipBackend -> ctypes -> libtnhip.so."""
import itertools

import numpy as np
import pytest

import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from oracle import numpy_oracle as orc
import cases as C

pytestmark = pytest.mark.gpu


def dev(hip, x, dtype=None):
  if dtype is ta.bfloat16:
    return hip.to_bfloat16(x)
  return hip.convert_to_tensor(np.asarray(x))


# ------------------------------------------------------------------ K1 permute
def test_permute_golden_bit_exact(hip, golden):
  for case in golden.cases["transpose"]:
    out = np.asarray(C.run_transpose(hip, golden, case))
    ref = golden[case["out"]]
    assert out.dtype == ref.dtype and out.shape == ref.shape
    assert out.tobytes() == ref.tobytes(), case


@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64, np.complex128, np.int32])
def test_permute_all_perms_rank4(hip, dtype):
  rng = np.random.default_rng(1)
  shape = (5, 1, 18, 33)
  x = rng.integers(-30000, 30000, size=shape).astype(dtype)
  d = dev(hip, x)
  for perm in itertools.permutations(range(4)):
    out = np.asarray(hip.transpose(d, perm))
    assert out.tobytes() == np.ascontiguousarray(np.transpose(x, perm)).tobytes(), perm


@pytest.mark.parametrize("shape,perm", [
    ((300, 200), (1, 0)), ((64, 48, 80), (2, 1, 0)), ((64, 48, 80), (1, 2, 0)), ((7, 130, 3, 70), (3, 2, 1, 0)),
    ((16,) * 6, (0, 2, 4, 1, 3, 5)), ((2, 512, 2, 512), (3, 1, 2, 0)), ((1000, 3), (1, 0)), ((3, 1000), (1, 0)),
    ((33, 65, 17), (0, 2, 1)), ((128, 128, 8), (1, 0, 2)),
    # full 64 x 128 tiles with 16-byte aligned rows: the 2-byte vector kernel (permute_tiled16)
    ((128, 256), (1, 0)), ((64, 128), (1, 0)), ((256, 4, 128), (2, 1, 0)), ((3, 64, 2, 128), (0, 3, 2, 1)),
    ((128, 3, 64), (2, 1, 0)),  # odd batch stride -> falls back to the scalar tiled kernel
    # round 6: destination-fastest extent 64 next to the source-fastest one (the chi = 64 MERA slices' pass): 64 x 64 tiles
    # of the 2-byte vector kernel (permute_tiled16_kernel<1, 64>)
    ((2, 3, 64, 5, 64), (0, 3, 1, 4, 2)), ((64, 64), (1, 0)), ((64, 8, 128), (1, 2, 0)), ((3, 64, 2, 192), (0, 2, 3, 1)),
    # 2-byte tensors whose (a, b) extents are >= 64 but not multiples of 64 / 128 (round 5: brick kernel instead of the
    # scalar 64 x 64 tiles -- the [K][N] -> [N][K] pass of a (96,)^4 tensor 1.48 -> 4.06 TB/s)
    ((96, 2, 3, 96), (1, 3, 2, 0)), ((72, 72), (1, 0)), ((100, 3, 100), (2, 1, 0)), ((160, 96), (1, 0)),
    ((66, 5, 130), (2, 1, 0)),
    # brick kernel, 16-byte vectors on both sides (runs of 144 / 1728 elements), on one side only, and not at all
    ((12, 12, 12, 12, 1, 12, 12, 12), (0, 3, 4, 5, 1, 6, 7, 2)), ((12,) * 6, (2, 1, 3, 4, 5, 0)),
    ((12,) * 6, (0, 1, 3, 5, 4, 2)), ((8,) * 7, (0, 4, 5, 1, 2, 6, 3)), ((6, 10, 12, 8, 12), (0, 3, 4, 1, 2)),
    ((5, 9, 12, 7, 12), (0, 3, 4, 1, 2)), ((16, 24, 8, 40), (2, 0, 3, 1)), ((10, 12, 14, 16), (3, 1, 0, 2)),
    # rows shorter than a 128-B line with the innermost dim preserved: the line-mate walk (gather2_kernel) with
    # g = 2, 4, 8 rows per line, a mate dim that g does not divide (g halves), and an odd one (plain gather)
    ((6, 16, 10, 16), (1, 0, 2, 3)), ((6, 12, 7, 8), (2, 0, 1, 3)), ((5, 8, 9, 4), (2, 1, 0, 3)),
    ((3, 6, 5, 8), (1, 0, 2, 3)), ((3, 7, 5, 8), (2, 0, 1, 3)), ((4, 4, 4, 4, 4, 16), (3, 1, 4, 0, 2, 5)),
])
@pytest.mark.parametrize("dtype", [np.float32, np.uint16, np.complex128])
def test_permute_tiled_and_gather_paths(hip, shape, perm, dtype):
  rng = np.random.default_rng(2)
  if dtype is np.uint16:
    x = rng.standard_normal(shape).astype(np.float32)
    d = hip.to_bfloat16(x)
    host = ta.round_to_bf16(x)
  else:
    x = (rng.standard_normal(shape) * 100).astype(dtype)
    d = dev(hip, x)
    host = x
  out = np.asarray(hip.transpose(d, perm))
  np.testing.assert_array_equal(out, np.transpose(host, perm))


def test_slice_diagonal_diagflat_bit_exact(hip):
  rng = np.random.default_rng(3)
  x = rng.standard_normal((6, 7, 8)).astype(np.float32)
  d = dev(hip, x)
  np.testing.assert_array_equal(np.asarray(hip.slice(d, (1, 2, 3), (4, 3, 2))), x[1:5, 2:5, 3:5])
  with pytest.raises(ValueError):
    hip.slice(d, (1, 2), (1, 1, 1))
  np.testing.assert_array_equal(np.asarray(d[2]), x[2])
  np.testing.assert_array_equal(np.asarray(d[:, ::2, -1]), x[:, ::2, -1])
  m = rng.standard_normal((3, 5, 6))
  dm = dev(hip, m)
  for off in (-2, 0, 1, 3):
    np.testing.assert_array_equal(np.asarray(hip.diagonal(dm, offset=off)), np.diagonal(m, off, -2, -1))
  np.testing.assert_array_equal(np.asarray(hip.diagonal(dm, axis1=0, axis2=2)), np.diagonal(m, 0, 0, 2))
  v = rng.standard_normal(5)
  for k in (0, 2, -1):
    np.testing.assert_array_equal(np.asarray(hip.diagflat(dev(hip, v), k)), np.diagflat(v, k))
  np.testing.assert_array_equal(np.asarray(hip.reshape(d, (-1, 8))), x.reshape(-1, 8))


# --------------------------------------------------------------------- K2 GEMM
def _gemm_case(hip, dtype, m, n, k, ta_, tb_, batch=None, variant="auto", rng=None):
  """tensordot/matmul through the backend in the requested storage layout."""
  rng = rng or np.random.default_rng(4)
  bshape = () if batch is None else (batch,)
  a_shape = bshape + ((k, m) if ta_ else (m, k))
  b_shape = bshape + ((n, k) if tb_ else (k, n))
  cplx = dtype in (np.complex64, np.complex128)
  a = rng.standard_normal(a_shape) + (1j * rng.standard_normal(a_shape) if cplx else 0)
  b = rng.standard_normal(b_shape) + (1j * rng.standard_normal(b_shape) if cplx else 0)
  if dtype is ta.bfloat16:
    a, b = orc.round_bf16(a), orc.round_bf16(b)
    da, db = hip.to_bfloat16(a), hip.to_bfloat16(b)
  else:
    a, b = a.astype(dtype), b.astype(dtype)
    da, db = dev(hip, a), dev(hip, b)
  _lib.check(hip.lib.tnh_gemm_set_variant(variant.encode()))
  try:
    if batch is None:
      out = hip.tensordot(da, db, [[0 if ta_ else 1], [1 if tb_ else 0]])
    else:
      ea = hip.transpose(da, (0, 2, 1)) if ta_ else da
      eb = hip.transpose(db, (0, 2, 1)) if tb_ else db
      out = hip.matmul(ea, eb)
    kernel = hip.lib.tnh_gemm_last_kernel().decode()
  finally:
    _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
  a64 = a.astype(np.complex128 if cplx else np.float64)
  b64 = b.astype(np.complex128 if cplx else np.float64)
  ea = np.swapaxes(a64, -1, -2) if ta_ else a64
  eb = np.swapaxes(b64, -1, -2) if tb_ else b64
  ref = ea @ eb
  return np.asarray(out), ref, kernel, np.sqrt(k)


@pytest.fixture(autouse=True)
def _kernel_tests_read_k_major_operands_in_place(request):
  """The tests of this file exercise the KERNELS: the in-place k-major readers included.  The backend's default
  policy (round 5) sends a compute-heavy product with a k-major operand through one K1 pass instead because that is
  faster (`HipBackend.kmajor_inplace_penalty`, tested in test_k_major_operands_take_the_faster_lowering); it is
  switched off here and restored afterwards."""
  if "hip" not in request.fixturenames:
    yield
    return
  be = request.getfixturevalue("hip")
  keep = be.kmajor_inplace_penalty
  be.kmajor_inplace_penalty = 0.0
  try:
    yield
  finally:
    be.kmajor_inplace_penalty = keep


GEMM_TOL = {np.float32: 3e-6, np.float64: 1e-14, np.complex64: 3e-6, np.complex128: 1e-14}


def assert_gemm(out, ref, dtype, k, scale_k=False, err_msg=""):
  """bf16 / f16 results: the parity rule of SURVEY 8c and of bench.verify_pair, elementwise and WITHOUT a sqrt(K)
  factor -- 2^-8 |ref| + 2^-10 rms(ref) (f16: 2^-11, 2^-13); see C.assert_half_gemm_close.  Other dtypes: the
  accumulate error of K terms in the storage precision (GEMM_TOL, optionally x sqrt(K) for the ragged sweep)."""
  if dtype is ta.bfloat16:
    return C.assert_half_gemm_close(out, ref, "bf16", err_msg)
  if dtype is np.float16:
    return C.assert_half_gemm_close(out, ref, "f16", err_msg)
  tol = GEMM_TOL[dtype] * (max(np.sqrt(k), 1.0) if scale_k else 1.0)
  return np.testing.assert_allclose(out, ref, rtol=tol, atol=tol * np.sqrt(k), err_msg=err_msg)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.float16, ta.bfloat16, np.complex64, np.complex128])
@pytest.mark.parametrize("ta_,tb_", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_all_layouts_ragged(hip, dtype, ta_, tb_):
  for (m, n, k) in [(1, 1, 1), (5, 7, 3), (33, 65, 17), (130, 70, 129), (64, 64, 64), (257, 129, 40)]:
    out, ref, kernel, sk = _gemm_case(hip, dtype, m, n, k, ta_, tb_)
    assert_gemm(out, ref, dtype, k, scale_k=True, err_msg=f"{kernel} {m}x{n}x{k}")


@pytest.mark.parametrize("variant", ["generic", "valu"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, ta.bfloat16])
def test_gemm_variants_agree(hip, variant, dtype):
  out, ref, kernel, sk = _gemm_case(hip, dtype, 200, 136, 96, 0, 1, variant=variant)
  assert_gemm(out, ref, dtype, 96, scale_k=True)
  assert ("valu" in kernel) == (variant == "valu")


@pytest.mark.parametrize("dtype", [np.float32, np.float64, ta.bfloat16])
def test_gemm_batched(hip, dtype):
  out, ref, _, sk = _gemm_case(hip, dtype, 40, 24, 56, 0, 0, batch=5)
  assert_gemm(out, ref, dtype, 56, scale_k=True)


@pytest.mark.parametrize("variant,expect", [("bf16_128", "bf16_nt_128x128x64"), ("bf16_256", "bf16_nt_256x256x64"),
                                            ("bf16_256pp", "bf16_nt_256x256x64_pp"),
                                            ("bf16_256pp:r1:p4", "bf16_nt_256x256x64_pp")])
@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 256, 128), (512, 384, 256), (200, 136, 192), (1024, 768, 512),
                                   (2048, 2304, 1088), (4096, 4096, 128)])
def test_gemm_bf16_speed_path(hip, variant, expect, dtype, m, n, k):
  """LDS-DMA + swizzled ds_read + 16x16x32 MFMA path; asymmetric random operands
  (catch row/col swaps), ragged M/N edges, both tile sizes."""
  out, ref, kernel, sk = _gemm_case(hip, dtype, m, n, k, 0, 1, variant=variant, rng=np.random.default_rng(m + n + k))
  assert kernel == expect
  assert_gemm(out, ref, dtype, k)


@pytest.mark.parametrize("ta_,tb_", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_gemm_f32_on_bf16_matrix_cores(hip, ta_, tb_):
  """Large f32 products run as ONE bf16 GEMM over the exact 3-way bf16 split of both operands (six
  products, smallest first): must be at least as close to float64 as the native f32 MFMA kernel, which
  ':s0' still selects.  Operands with 6 decades of dynamic range; K % 64 != 0 exercises the zero fill."""
  m, n, k = 3584, 3840, 1096
  rng = np.random.default_rng(17 + 2 * ta_ + tb_)
  a_shape, b_shape = ((k, m) if ta_ else (m, k)), ((n, k) if tb_ else (k, n))
  a = (rng.standard_normal(a_shape) * np.exp(rng.uniform(-7, 7, a_shape))).astype(np.float32)
  b = rng.standard_normal(b_shape).astype(np.float32)
  da, db = dev(hip, a), dev(hip, b)
  axes = [[0 if ta_ else 1], [1 if tb_ else 0]]
  a64 = (a.T if ta_ else a).astype(np.float64)
  b64 = (b.T if tb_ else b).astype(np.float64)
  exact, scale = a64 @ b64, np.abs(a64) @ np.abs(b64)
  err = {}
  for variant in ("auto", "auto:s0"):
    _lib.check(hip.lib.tnh_gemm_set_variant(variant.encode()))
    try:
      out = np.asarray(hip.tensordot(da, db, axes))
      kernel = hip.lib.tnh_gemm_last_kernel().decode()
    finally:
      _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
    assert out.dtype == np.float32
    err[kernel] = float((np.abs(out - exact) / scale).max())
  native = [kname for kname in err if kname.startswith("mfma_f32_128x128")]
  assert set(err) == {"f32_as_3xbf16_nt_256x256x64_pp", native[0]} and len(native) == 1, err
  assert err["f32_as_3xbf16_nt_256x256x64_pp"] <= 5e-7                      # ~4 eps_f32 of sum |a||b| at K ~ 1100
  assert err["f32_as_3xbf16_nt_256x256x64_pp"] <= 1.5 * err[native[0]]


def test_gemm_f32_split_guard_matches_numpy_on_inf_and_tiny_inputs(hip):
  """the 3 x bf16 split cannot carry inf (inf - inf -> NaN rows), values beyond the bf16 range
  or near-subnormal values.  Its split kernels flag such operands and a predicated launch of the f32 MFMA kernel
  recomputes the product: the result is NumPy's (+-inf where NumPy has +-inf, no stray NaN, tiny values kept)."""
  m, n, k = 3584, 3840, 1024
  rng = np.random.default_rng(5)
  a = rng.standard_normal((m, k)).astype(np.float32)
  b = rng.standard_normal((k, n)).astype(np.float32)
  with np.errstate(all="ignore"):
    for case in ("inf", "big", "tiny", "clean"):
      aa, bb = a.copy(), b.copy()
      if case == "inf":
        aa[7, 11] = np.inf
        bb[3, 5] = -np.inf
      elif case == "big":
        aa[9, 1] = 3.40e38            # finite in f32, beyond the largest bf16
        bb[1, :] = 0.0
        bb[1, 4] = 0.5
      elif case == "tiny":
        aa *= np.float32(1e-36)       # products ~1e-36: the lo parts of the split would be subnormal
      out = np.asarray(hip.tensordot(dev(hip, aa), dev(hip, bb), [[1], [0]]))
      assert hip.lib.tnh_gemm_last_kernel().decode() == "f32_as_3xbf16_nt_256x256x64_pp"
      ref = aa.astype(np.float64) @ bb.astype(np.float64)
      ref32 = ref.astype(np.float32)
      # same non-finite pattern as NumPy's float32 product
      np.testing.assert_array_equal(np.isnan(out), np.isnan(aa @ bb), err_msg=case)
      np.testing.assert_array_equal(np.isposinf(out), np.isposinf(aa @ bb), err_msg=case)
      np.testing.assert_array_equal(np.isneginf(out), np.isneginf(aa @ bb), err_msg=case)
      fin = np.isfinite(ref32) & np.isfinite(out)
      scale = np.abs(aa.astype(np.float64)) @ np.abs(bb.astype(np.float64))
      scale[~np.isfinite(scale)] = np.inf
      assert np.all(np.abs(out[fin] - ref[fin]) <= 2e-6 * scale[fin] + 1e-44), case


def test_gemm_f32_split_only_for_large_products(hip):
  out, ref, kernel, sk = _gemm_case(hip, np.float32, 1024, 1024, 2048, 0, 1)
  # 16 tiles of 256^2: the native f32 kernels (from round 6 on through the mid-K split: 64 tiles of 128^2 <= CUs / 2),
  # not the 3 x bf16 split
  assert kernel == "splitk" or kernel.startswith("mfma_f32_128x128"), kernel
  np.testing.assert_allclose(out, ref, rtol=GEMM_TOL[np.float32] * sk, atol=GEMM_TOL[np.float32] * sk * np.sqrt(2048))


@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (512, 768, 192), (1024, 768, 448), (2048, 2304, 1088)])
def test_gemm_bf16_four_wave_variant(hip, dtype, m, n, k):
  """A/B variant ':p6' (4 waves x 128x128 wave tiles, hand-scheduled asm MFMA / ds_read stream, 2 x 64 KiB
  dynamic LDS): same tolerance as the default speed path, and the odd / even K-tile counts of its pipeline."""
  out, ref, kernel, sk = _gemm_case(hip, dtype, m, n, k, 0, 1, variant="bf16_256pp:p6",
                                    rng=np.random.default_rng(m + n + k))
  assert kernel == "bf16_nt_256x256x64_w4"
  assert_gemm(out, ref, dtype, k)
  out2, _, kernel2, _ = _gemm_case(hip, dtype, m, n, k, 0, 1, variant="bf16_256pp", rng=np.random.default_rng(m + n + k))
  assert kernel2 == "bf16_nt_256x256x64_pp"
  np.testing.assert_array_equal(out, out2)          # same accumulation order: bit-identical


@pytest.mark.parametrize("variant,expect", [("bf16_ragged", "bf16_nt_ragged_"), ("bf16_ragged_128x128", "bf16_nt_ragged_128x128x64"),
                                            ("bf16_ragged_64x256", "bf16_nt_ragged_64x256x64"),
                                            ("bf16_ragged_256x64", "bf16_nt_ragged_256x64x64"),
                                            ("bf16_ragged:p1", "bf16_nt_ragged_"), ("bf16_ragged:r2", "bf16_nt_ragged_")])
@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
def test_gemm_bf16_ragged_path(hip, variant, expect, dtype):
  """Register-staged matrix-core kernel: any M, N, K (K tail zero-filled, k-steps past K
  skipped), every row alignment (K = 12 -> 8-B loads, 6 -> 4-B, 7 -> 2-B), scalar C stores
  when N % 4 != 0, all three tile shapes."""
  for (m, n, k) in [(1, 1, 1), (5, 7, 3), (33, 65, 17), (130, 70, 129), (144, 300, 144), (12, 1000, 12),
                    (257, 129, 40), (200, 136, 192), (64, 64, 64), (300, 20, 100), (1, 1, 1728), (144, 1728, 6),
                    (20, 36, 7), (513, 258, 72), (70, 264, 33), (131, 520, 200)]:
    out, ref, kernel, sk = _gemm_case(hip, dtype, m, n, k, 0, 1, variant=variant, rng=np.random.default_rng(m + n + k))
    assert kernel.startswith(expect), kernel
    assert_gemm(out, ref, dtype, k, err_msg=f"{kernel} {m}x{n}x{k}")


def test_gemm_bf16_ragged_auto_dispatch_and_batch(hip):
  """Auto dispatch: NT products that break the LDS-DMA alignment rules take the ragged matrix-core
  kernel (not the f32-MFMA fallback); strided batches with odd strides included."""
  rng = np.random.default_rng(12)
  a = orc.round_bf16(rng.standard_normal((144, 144)))
  b = orc.round_bf16(rng.standard_normal((5000, 144)))
  out = hip.tensordot(hip.to_bfloat16(a), hip.to_bfloat16(b), [[1], [1]])
  assert hip.lib.tnh_gemm_last_kernel().decode() == "bf16_nt_ragged_128x128x64_smallk"
  C.assert_half_gemm_close(np.asarray(out), a @ b.T, "bf16")
  # batched NT through the C ABI: batch 3, K = 10 (4-B loads), strides not multiples of 8
  import ctypes
  from tensornetwork_amd.device_tensor import DeviceTensor
  A = orc.round_bf16(rng.standard_normal((3, 37, 10)))
  B = orc.round_bf16(rng.standard_normal((3, 21, 10)))
  da, db = hip.to_bfloat16(A), hip.to_bfloat16(B)
  c = DeviceTensor.empty((3, 37, 21), _lib.F32)
  _lib.check(hip.lib.tnh_gemm(_lib.BF16, _lib.F32, 0, 1, 37, 21, 10, ctypes.c_void_p(da.ptr), 10,
                              ctypes.c_void_p(db.ptr), 10, ctypes.c_void_p(c.ptr), 21, 3, 370, 210, 37 * 21))
  assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_nt_ragged")
  np.testing.assert_allclose(np.asarray(c), np.einsum("bmk,bnk->bmn", A, B), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
def test_gemm_stream_small_times_long(hip, dtype):
  """Small x very long NT products take the streaming kernel (small operand resident in LDS, long operand
  prefetched through registers) in both orientations; same MFMA sequence as the ragged tile kernel, so the
  two agree bit for bit.  Ragged long side (remainder through the tile kernel), K tails (8, 16, 56, 104 ->
  zero-filled k-steps); shapes that break the 16-byte rules (K = 7, K = 100 with N odd) take the tile kernels."""
  for (m, n, k) in [(144, 70000, 144), (70000, 144, 144), (16, 65536, 8), (192, 66008, 192), (66001, 192, 192),
                    (72, 65544, 104), (65543, 80, 16), (144, 65600, 56), (65600, 32, 8), (65600, 33, 7), (50, 65543, 100),
                    (100, 66001, 64), (176, 65536, 128)]:
    rng = np.random.default_rng(m + n + k)
    out, ref, kernel, _ = _gemm_case(hip, dtype, m, n, k, 0, 1, rng=rng)
    # 16-byte rules (K % 8, ldc = n % 8) and the size gate (short side 65 .. 192, K >= 16); the rest: tile kernels
    streamed = (k % 8 == 0 and n % 8 == 0 and k >= 16 and 64 < min(m, n) <= 192)
    assert kernel.startswith("bf16_nt_stream") == streamed, (kernel, m, n, k)
    rng = np.random.default_rng(m + n + k)
    tiled, _, kernel2, _ = _gemm_case(hip, dtype, m, n, k, 0, 1, variant="bf16_ragged_128x128", rng=rng)
    assert kernel2.startswith("bf16_nt_ragged"), kernel2
    np.testing.assert_array_equal(out, tiled, err_msg=f"{kernel} vs {kernel2} {m}x{n}x{k}")
    assert_gemm(out, ref, dtype, k, err_msg=f"{kernel} {m}x{n}x{k}")


def test_gemm_stream_padded_ldc_and_forced_variant(hip):
  """C rows padded (ldc > N) in both orientations: the padding stays untouched; the forced variant refuses
  shapes outside its range."""
  import ctypes
  rng = np.random.default_rng(21)
  for (m, n, k, ldc) in [(70000, 104, 40, 112), (100, 70001, 40, 70008)]:
    A = orc.round_bf16(rng.standard_normal((m, k)))
    B = orc.round_bf16(rng.standard_normal((n, k)))
    c = hip.to_bfloat16(np.full((m, ldc), 7.0, dtype=np.float32))
    _lib.check(hip.lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, ctypes.c_void_p(hip.to_bfloat16(A).ptr), k,
                                ctypes.c_void_p(hip.to_bfloat16(B).ptr), k, ctypes.c_void_p(c.ptr), ldc, 1, 0, 0, 0))
    assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_nt_stream")
    got = np.asarray(c)
    C.assert_half_gemm_close(got[:, :n], A @ B.T, "bf16")
    np.testing.assert_array_equal(got[:, n:], 7.0)
  _lib.check(hip.lib.tnh_gemm_set_variant(b"bf16_stream"))
  try:
    a = hip.to_bfloat16(rng.standard_normal((300, 64)))
    b = hip.to_bfloat16(rng.standard_normal((400, 64)))
    c = hip.to_bfloat16(np.zeros((300, 400), dtype=np.float32))
    st = hip.lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, 300, 400, 64, ctypes.c_void_p(a.ptr), 64, ctypes.c_void_p(b.ptr), 64,
                          ctypes.c_void_p(c.ptr), 400, 1, 0, 0, 0)
    assert st == _lib.ERR_UNSUPPORTED, st          # both sides > 192
    a = hip.to_bfloat16(rng.standard_normal((100, 64)))
    out = hip.tensordot(a, b, [[1], [1]])          # forced on a short product: still correct
    assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_nt_stream")
    C.assert_half_gemm_close(np.asarray(out), np.asarray(a).astype(np.float64) @ np.asarray(b).astype(np.float64).T, "bf16")
  finally:
    _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))


def test_gemm_bf16_ragged_padded_ldc(hip):
  """C rows padded to ldc > N (N % 8 != 0, ldc % 8 == 0): LDS-staged epilogue with a partial last chunk;
  the padding columns must stay untouched."""
  import ctypes
  from tensornetwork_amd.device_tensor import DeviceTensor
  rng = np.random.default_rng(13)
  m, n, k, ldc = 150, 21, 40, 24
  A = orc.round_bf16(rng.standard_normal((m, k)))
  B = orc.round_bf16(rng.standard_normal((n, k)))
  c = hip.to_bfloat16(np.full((m, ldc), 7.0, dtype=np.float32))
  _lib.check(hip.lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, ctypes.c_void_p(hip.to_bfloat16(A).ptr), k,
                              ctypes.c_void_p(hip.to_bfloat16(B).ptr), k, ctypes.c_void_p(c.ptr), ldc, 1, 0, 0, 0))
  assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_nt_ragged")
  got = np.asarray(c)
  C.assert_half_gemm_close(got[:, :n], A @ B.T, "bf16")
  np.testing.assert_array_equal(got[:, n:], 7.0)


def test_gemm_bf16_fp32_output_is_tighter(hip):
  rng = np.random.default_rng(9)
  a = orc.round_bf16(rng.standard_normal((256, 512)))
  b = orc.round_bf16(rng.standard_normal((384, 512)))
  be32 = ta.HipBackend(half_output="float32")
  out = be32.tensordot(hip.to_bfloat16(a), hip.to_bfloat16(b), [[1], [1]])
  assert out.dtype == np.float32
  assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_nt")
  ref = a.astype(np.float64) @ b.astype(np.float64).T
  np.testing.assert_allclose(np.asarray(out), ref, rtol=2e-5, atol=2e-5 * np.sqrt(512))


def test_gemm_bf16_auto_dispatch_uses_speed_path(hip):
  rng = np.random.default_rng(10)
  a = hip.to_bfloat16(rng.standard_normal((512, 256)))
  b = hip.to_bfloat16(rng.standard_normal((256, 512)))   # KN storage: host permutes once, then NT kernel
  out = hip.tensordot(a, b, 1)
  assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_nt")
  ref = np.asarray(a).astype(np.float64) @ np.asarray(b).astype(np.float64)
  C.assert_half_gemm_close(np.asarray(out), ref, "bf16")


@pytest.mark.parametrize("dtype,kernel,tol", [(np.complex64, "mfma_f32_128x128", 3e-6), (np.complex128, "mfma_f64_64x64x16", 1e-14)])
@pytest.mark.parametrize("ta_,tb_", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_complex_on_matrix_cores(hip, dtype, kernel, tol, ta_, tb_):
  """complex64 / complex128 contraction = one real MFMA GEMM on the interleaved images of A and C
  against the 2x2-block real expansion of B (tnh_complex_expand), every storage layout."""
  for (m, n, k) in [(200, 136, 96), (64, 257, 33), (512, 384, 256)]:
    out, ref, name, sk = _gemm_case(hip, dtype, m, n, k, ta_, tb_, rng=np.random.default_rng(m + n + k))
    # (the 2 k = 512-deep real product of the last shape is in the mid-K split-K regime of round 4: same kernels,
    # K slices, partials summed in a fixed order)
    assert name.startswith(kernel) or (name == "splitk" and 2 * k >= 512), name
    np.testing.assert_allclose(out, ref, rtol=tol * sk * 4, atol=tol * k * 2)


# tolerance ladder of the reference's own property test (backends/tensorflow/tensordot2_test.py:143-159):
# fp16 0.05, fp32 / complex64 1e-5, else 1e-12 -- bf16 gets 2^-6 (one bit less mantissa than fp16's 0.05 class)
_TD_TOL = {np.float16: 5e-2, ta.bfloat16: 8e-2, np.float32: 1e-5, np.complex64: 1e-5, np.float64: 1e-12,
           np.complex128: 1e-12}


@pytest.mark.parametrize("dtype", [np.float16, ta.bfloat16, np.float32, np.float64, np.complex64, np.complex128])
def test_tensordot_random_axes_property(hip, dtype):
  """tensordot2_test.py:163-207 restated for the hip backend: random ranks (1-4), dims (1-9, plus a few
  larger ones so the matrix-core kernels and every layout branch of the lowering are hit), random
  axes subsets and scalar `axes`, against np.tensordot on the same (rounded) inputs."""
  rng = np.random.default_rng(hash(str(dtype)) % 1000)
  tol = _TD_TOL[dtype]
  cplx = dtype in (np.complex64, np.complex128)
  for trial in range(60):
    big = trial % 5 == 0
    rank_a, rank_b = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    nc = int(rng.integers(0, min(rank_a, rank_b) + 1))
    hi = 40 if big else 10
    cdims = [int(rng.integers(1, hi)) for _ in range(nc)]
    shape_a = [int(rng.integers(1, hi)) for _ in range(rank_a)]
    shape_b = [int(rng.integers(1, hi)) for _ in range(rank_b)]
    axes_a = sorted(rng.choice(rank_a, nc, replace=False).tolist())
    axes_b = rng.choice(rank_b, nc, replace=False).tolist()
    rng.shuffle(axes_a)
    for d, (xa, xb) in zip(cdims, zip(axes_a, axes_b)):
      shape_a[xa] = d
      shape_b[xb] = d
    free = int(np.prod(shape_a)) // max(int(np.prod(cdims)), 1) * (int(np.prod(shape_b)) // max(int(np.prod(cdims)), 1))
    if max(int(np.prod(shape_a)), int(np.prod(shape_b)), free) > 300_000:
      continue   # keep every operand and the result small: this is a parity test, not a stress test
    a = rng.standard_normal(shape_a) + (1j * rng.standard_normal(shape_a) if cplx else 0)
    b = rng.standard_normal(shape_b) + (1j * rng.standard_normal(shape_b) if cplx else 0)
    if dtype is ta.bfloat16:
      a, b = orc.round_bf16(a), orc.round_bf16(b)
      da, db = hip.to_bfloat16(a), hip.to_bfloat16(b)
    else:
      a, b = a.astype(dtype), b.astype(dtype)
      da, db = dev(hip, a), dev(hip, b)
    ref = np.tensordot(a.astype(np.complex128 if cplx else np.float64), b.astype(np.complex128 if cplx else np.float64),
                       [axes_a, axes_b])
    out = np.asarray(hip.tensordot(da, db, [axes_a, axes_b]))
    k = max(int(np.prod(cdims)), 1)
    assert out.shape == ref.shape, (shape_a, shape_b, axes_a, axes_b)
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol * np.sqrt(k) * 4,
                               err_msg=f"{shape_a} {shape_b} {axes_a} {axes_b} {hip.lib.tnh_gemm_last_kernel().decode()}")
  # scalar axes (tensordot2_test.py:163-184): last n of a with first n of b
  for n in (0, 1, 2):
    a = rng.standard_normal((3, 4, 5)).astype(np.float32 if dtype is ta.bfloat16 else dtype)
    b = rng.standard_normal((4, 5, 6)[2 - n:] + (7,) if n else (2, 3)).astype(a.dtype)
    if n:
      b = rng.standard_normal(a.shape[3 - n:] + (7,)).astype(a.dtype)
    out = np.asarray(hip.tensordot(dev(hip, a), dev(hip, b), n))
    np.testing.assert_allclose(out, np.tensordot(a.astype(np.float64) if not cplx else a, b.astype(np.float64) if not cplx else b, n),
                               rtol=max(tol, 1e-5), atol=max(tol, 1e-5) * 10)


@pytest.mark.parametrize("ta_,tb_", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_f64_128_tile(hip, ta_, tb_):
  """f64 products with >= 128 tiles of 128x128 take the wider MFMA kernel; ragged edges, every layout."""
  out, ref, kernel, sk = _gemm_case(hip, np.float64, 1500, 1400, 203, ta_, tb_, rng=np.random.default_rng(3))
  assert kernel.startswith("mfma_f64_128x128"), kernel    # x32 (odd leading dimension) or x16_v2 (16-byte loads)
  np.testing.assert_allclose(out, ref, rtol=1e-14 * sk, atol=1e-14 * 203)
  out, ref, kernel, sk = _gemm_case(hip, np.complex128, 1400, 800, 160, ta_, tb_, rng=np.random.default_rng(4))
  assert kernel.startswith("mfma_f64_128x128"), kernel    # complex128 rides the same kernels (real expansion)
  np.testing.assert_allclose(out, ref, rtol=1e-14 * sk * 4, atol=1e-14 * 160 * 2)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, ta.bfloat16, np.float16])
def test_gemm_split_k_small_output(hip, dtype):
  """Few output tiles + long K (inner products, small environments): K is split over workgroups and the
  f32 / f64 partials are summed by the reduction kernel; every layout, ragged K tail."""
  rng = np.random.default_rng(7)
  for (m, n, k, ta_, tb_) in [(1, 1, 262144, 0, 1), (3, 5, 70001, 0, 0), (64, 130, 20000, 1, 1), (200, 100, 9000, 1, 0)]:
    out, ref, kernel, sk = _gemm_case(hip, dtype, m, n, k, ta_, tb_, rng=rng)
    tiny = m <= 4 and n <= 4       # round 6: one workgroup (gemm_tiny_kernel; bf16 / f16 too since the last session)
    assert kernel == ("tiny_1wg" if tiny else "splitk"), (kernel, m, n, k)
    if dtype in (ta.bfloat16, np.float16):      # split-K partials are f32, rounded once at the end: the same rule
      assert_gemm(out, ref, dtype, k, err_msg=f"{m}x{n}x{k}")
      continue
    tol = {np.float32: 2e-6, np.float64: 1e-14}[dtype]
    np.testing.assert_allclose(out, ref, rtol=tol * 4, atol=tol * np.sqrt(k) * 4, err_msg=f"{m}x{n}x{k}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gemm_split_k_mid_size_f32_f64(hip, dtype):
  """Round 4: f32 / f64 products with few tiles and K of a few hundred to a few thousand (MPS transfer-matrix steps:
  (512 ... 1024)^3 / 2) are split over K too -- one 128 x 128 f32 tile keeps a CU busy for K x 0.054 us; below the
  thresholds (small K, tiny outputs, many tiles) nothing changes.  Every layout, ragged K."""
  rng = np.random.default_rng(11)
  kmin = 512
  for (m, n, k, ta_, tb_, want) in [(512, 512, 1024, 0, 0, "splitk"), (1024, 512, 1000, 0, 1, "splitk"), (512, 1024, kmin, 1, 0, "splitk"),
                                    (200, 300, 2500, 1, 1, "splitk"), (512, 512, kmin - 64, 0, 0, None),
                                    # round 6: one side <= 128 (MPS site legs against a bond) -- one launch, a workgroup per 16 x 16 tile
                                    (32, 512, 2048, 0, 0, "skinny_16x16"), (2, 1024, 512, 0, 0, "skinny_16x16"),
                                    (4, 1024, 512, 0, 1, "skinny_16x16"), (1024, 8, 640, 1, 1, "skinny_16x16"),
                                    (40, 512, 8192, 0, 0, "splitk"),     # ... beyond K = 4096 the split-K path (slices of >= 64)
                                    (2048, 2048, 1024, 0, 1, None)]:
    out, ref, kernel, sk = _gemm_case(hip, dtype, m, n, k, ta_, tb_, rng=rng)
    assert (kernel == want) if want else (kernel not in ("splitk", "skinny_16x16")), (kernel, m, n, k)
    tol = 2e-6 if dtype == np.float32 else 1e-14
    np.testing.assert_allclose(out, ref, rtol=tol * 4, atol=tol * np.sqrt(k) * 4, err_msg=f"{m}x{n}x{k}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("ta_,tb_", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_skinny_one_workgroup_per_16x16_tile(hip, dtype, ta_, tb_):
  """Round 6 (gemm_skinny_kernel): f32 / f64 products with one side <= 128, K = 256 ... 4096: every storage form, ragged
  M / N / K (K % 16 != 0: masked tail; rows past the edge re-read and never stored), sixteen- and four-wave forms, and
  the fall-back when a k-contiguous operand's rows are not 16-byte aligned."""
  rng = np.random.default_rng(41 + 2 * ta_ + tb_)
  for (m, n, k) in [(2, 1024, 512), (17, 1000, 300), (128, 2048, 4096), (100, 33, 257), (1, 700, 256), (64, 1536, 1040)]:
    out, ref, kernel, _ = _gemm_case(hip, dtype, m, n, k, ta_, tb_, rng=rng)
    aligned = (ta_ or k % (4 if dtype == np.float32 else 2) == 0) and (not tb_ or k % (4 if dtype == np.float32 else 2) == 0)
    assert (kernel == "skinny_16x16") == aligned, (kernel, m, n, k)
    tol = 2e-6 if dtype == np.float32 else 1e-14
    np.testing.assert_allclose(out, ref, rtol=tol * 4, atol=tol * np.sqrt(k) * 4, err_msg=f"{kernel} {m}x{n}x{k}")


def test_tensordot_golden(hip, golden):
  for case in golden.cases["tensordot"]:
    C.assert_close(C.run_tensordot(hip, golden, case), golden[case["out"]])


def test_tensordot_errors_and_empty(hip):
  a, b = dev(hip, np.ones((2, 3))), dev(hip, np.ones((3, 2)))
  with pytest.raises(ValueError, match="shape-mismatch for sum"):
    hip.tensordot(a, b, [[0, 1], [0]])
  with pytest.raises(ValueError, match="shape-mismatch for sum"):
    hip.tensordot(a, b, [[0], [0]])
  z = hip.tensordot(dev(hip, np.ones((4, 0))), dev(hip, np.ones((0, 5))), 1)
  np.testing.assert_array_equal(np.asarray(z), np.zeros((4, 5)))
  e = hip.tensordot(dev(hip, np.ones((0, 3))), dev(hip, np.ones((3, 5))), 1)
  assert e.shape == (0, 5)
  with pytest.raises(ValueError):
    hip.matmul(dev(hip, np.ones(3)), dev(hip, np.ones((3, 3))))
  with pytest.raises(TypeError):
    hip.convert_to_tensor([1, 2, 3])


# --------------------------------------------------------- K3-K6 helper kernels
def test_misc_golden(hip, golden):
  for case in golden.cases["misc"]:
    res = C.run_misc(hip, golden, case)
    for name, val in res.items():
      C.assert_close(val, golden[case[name]])
    x = dev(hip, golden[case["x"]])
    v = dev(hip, golden[case["v"]])
    C.assert_close(hip.norm(x), golden[case["norm"]])
    C.assert_close(hip.sqrt(hip.abs(x)), golden[case["sqrtabs"]])
    C.assert_close(hip.subtraction(x, v), golden[case["sub"]])
    C.assert_close(hip.divide(x, v), golden[case["div"]])
    C.assert_close(hip.diagonal(dev(hip, golden[case["m"]])), golden[case["diagonal"]])


@pytest.mark.parametrize("dtype", [np.float32, np.float64, ta.bfloat16, np.complex128])
def test_reductions_large(hip, dtype):
  rng = np.random.default_rng(5)
  x = rng.standard_normal((37, 1000, 19))
  if dtype is np.complex128:
    x = x + 1j * rng.standard_normal(x.shape)
  if dtype is ta.bfloat16:
    x = orc.round_bf16(x)
    d = hip.to_bfloat16(x)
    tol = dict(rtol=1e-2, atol=0.5)
  else:
    x = x.astype(dtype)
    d = dev(hip, x)
    tol = dict(rtol=1e-4, atol=1e-3) if dtype is np.float32 else dict(rtol=1e-12, atol=1e-10)
  x64 = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)
  for axis in [(0,), (1,), (2,), (0, 1), (1, 2), (0, 2), (0, 1, 2)]:
    np.testing.assert_allclose(np.asarray(hip.sum(d, axis)), x64.sum(axis), **tol)
  np.testing.assert_allclose(np.asarray(hip.sum(d, (1,), keepdims=True)), x64.sum(1, keepdims=True), **tol)
  np.testing.assert_allclose(np.asarray(hip.norm(d)), np.linalg.norm(x64), rtol=tol["rtol"])
  big = rng.standard_normal(3_000_001).astype(np.float32)
  np.testing.assert_allclose(np.asarray(hip.sum(dev(hip, big), (0,))), big.astype(np.float64).sum(), rtol=1e-5, atol=1e-2)
  t = rng.standard_normal((3, 2050, 2050)).astype(np.float32)
  np.testing.assert_allclose(np.asarray(hip.trace(dev(hip, t))), np.trace(t.astype(np.float64), axis1=-2, axis2=-1),
                             rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_elementwise_math(hip, dtype):
  rng = np.random.default_rng(6)
  x = rng.standard_normal((11, 13)) + 0.5
  if np.dtype(dtype).kind == "c":
    x = x + 1j * rng.standard_normal(x.shape)
  x = x.astype(dtype)
  d = dev(hip, x)
  rt = 3e-6 if np.dtype(dtype).itemsize in (4, 8) and np.dtype(dtype) != np.float64 else 1e-13
  for name, fn in [("sqrt", np.sqrt), ("exp", np.exp), ("log", np.log), ("sin", np.sin), ("cos", np.cos),
                   ("abs", np.abs), ("sign", np.sign), ("conj", np.conj)]:
    with np.errstate(all="ignore"):
      ref = fn(x.astype(np.complex128) if np.dtype(dtype).kind == "c" else np.abs(x.astype(np.float64))
               if name in ("sqrt", "log") else x.astype(np.float64))
    arg = d if not (name in ("sqrt", "log") and np.dtype(dtype).kind != "c") else hip.abs(d)
    np.testing.assert_allclose(np.asarray(getattr(hip, name)(arg)), ref, rtol=rt, atol=rt, err_msg=name)
  np.testing.assert_allclose(np.asarray(d * 2.5 - d / 4 + 1.0), x * 2.5 - x / 4 + 1.0, rtol=rt, atol=rt)
  np.testing.assert_allclose(np.asarray(3.0 / (d + 5.0)), 3.0 / (x + 5.0), rtol=rt, atol=rt)
  np.testing.assert_allclose(np.asarray(hip.power(hip.abs(d), 1.5)), np.abs(x) ** 1.5, rtol=10 * rt, atol=rt)
  y = rng.standard_normal((13,)).astype(dtype)
  np.testing.assert_allclose(np.asarray(d + dev(hip, y)), x + y, rtol=rt, atol=rt)
  np.testing.assert_allclose(np.asarray(dev(hip, y.reshape(1, 13)) * d), y * x, rtol=rt, atol=rt)
  with pytest.raises(ValueError):
    hip.addition(d, dev(hip, np.ones((4, 4), dtype=dtype)))
  with pytest.raises(ValueError):
    hip.broadcast_right_multiplication(d, d)
  with pytest.raises(ValueError):
    hip.broadcast_left_multiplication(d, d)


def test_init_functions(hip):
  np.testing.assert_array_equal(np.asarray(hip.eye(4, dtype=np.float32, M=6)), np.eye(4, 6, dtype=np.float32))
  np.testing.assert_array_equal(np.asarray(hip.ones((2, 3), np.complex128)), np.ones((2, 3), np.complex128))
  np.testing.assert_array_equal(np.asarray(hip.zeros((2, 3))), np.zeros((2, 3)))
  a = np.asarray(hip.randn((4, 5), dtype=np.float64, seed=10))
  np.random.seed(10)
  np.testing.assert_array_equal(a, np.random.randn(4, 5))  # same stream as the NumPy backend
  u = np.asarray(hip.random_uniform((100,), (-2, 3), dtype=np.float32, seed=3))
  assert u.min() >= -2 and u.max() <= 3
  s = hip.serialize_tensor(dev(hip, a))
  np.testing.assert_array_equal(np.asarray(hip.deserialize_tensor(s)), a)
  assert hip.item(hip.convert_to_tensor(3.5)) == 3.5
  assert hip.eps(np.float32) == np.finfo(np.float32).eps


def test_casts(hip):
  rng = np.random.default_rng(7)
  x = rng.standard_normal(1000).astype(np.float32)
  d = dev(hip, x)
  np.testing.assert_array_equal(np.asarray(hip.cast(d, ta.bfloat16)), orc.round_bf16(x))  # device RNE == host RNE
  np.testing.assert_array_equal(np.asarray(hip.cast(d, np.float16)), x.astype(np.float16))
  np.testing.assert_array_equal(np.asarray(hip.cast(d, np.float64)), x.astype(np.float64))
  np.testing.assert_array_equal(np.asarray(hip.cast(d, np.complex64)), x.astype(np.complex64))


# --------------------------------------------------------------------- K7 SVD
def _check_svd(hip, a, rtol_s, otol, **kw):
  u, s, vh, rest = hip.svd(dev(hip, a), 1, **kw)
  u, s, vh, rest = (np.asarray(t) for t in (u, s, vh, rest))
  ur, sr, vhr, restr = orc.svd(a.astype(np.float64), 1, **kw)
  s0 = max(float(sr[0]) if sr.size else 0.0, float(np.abs(a).max()), 1e-30)
  assert s.shape == sr.shape and rest.shape == restr.shape and u.shape == ur.shape and vh.shape == vhr.shape
  np.testing.assert_allclose(s, sr, rtol=0, atol=rtol_s * s0)
  np.testing.assert_allclose(rest, restr, rtol=0, atol=rtol_s * s0)
  k = s.size
  if k:
    np.testing.assert_allclose(u.T @ u, np.eye(k), atol=otol)
    np.testing.assert_allclose(vh @ vh.T, np.eye(k), atol=otol)
    approx = (u * s) @ vh
    best = (ur * sr) @ vhr
    assert np.linalg.norm(approx - best) <= 10 * otol * max(np.linalg.norm(a), 1e-30)
  return u, s, vh, rest


@pytest.mark.parametrize("dtype,rtol_s,otol", [(np.float32, 1e-5, 1e-4), (np.float64, 1e-13, 1e-12)])
@pytest.mark.parametrize("shape", [(1, 1), (2, 9), (9, 2), (17, 17), (64, 48), (48, 64), (130, 131), (257, 40)])
def test_svd_random(hip, dtype, rtol_s, otol, shape):
  rng = np.random.default_rng(shape[0] * 1000 + shape[1])
  a = rng.standard_normal(shape).astype(dtype)
  _check_svd(hip, a, rtol_s, otol)
  _check_svd(hip, a, rtol_s, otol, max_singular_values=max(1, min(shape) // 3))
  _check_svd(hip, a, rtol_s, otol, max_truncation_error=0.3, relative=True)


@pytest.mark.parametrize("dtype,rtol_s,otol", [(np.float32, 1e-5, 1e-4), (np.float64, 1e-13, 1e-12)])
def test_svd_rank_deficient_and_zero(hip, dtype, rtol_s, otol):
  rng = np.random.default_rng(8)
  b = (rng.standard_normal((40, 6)) @ rng.standard_normal((6, 30))).astype(dtype)
  b[:, 3] = 0
  b[7, :] = 0
  _check_svd(hip, b, 20 * rtol_s, 20 * otol)
  z = np.zeros((12, 20), dtype=dtype)
  u, s, vh, _ = _check_svd(hip, z, rtol_s, otol)
  assert np.all(s == 0)
  d = np.diag([2.0, 1.0, 0.2, 0.1]).astype(dtype)  # decompositions_test.py:92-108
  _, _, _, trunc = hip.svd(dev(hip, d), 1, max_truncation_error=0.2, relative=False)
  np.testing.assert_allclose(np.asarray(trunc), [0.1], rtol=1e-5)
  _, _, _, trunc = hip.svd(dev(hip, d), 1, max_truncation_error=0.2, relative=True)
  np.testing.assert_allclose(np.asarray(trunc), [0.2, 0.1], rtol=1e-5)


@pytest.mark.parametrize("dtype,rtol_s,otol", [(np.float32, 1e-5, 1e-4), (np.float64, 1e-13, 1e-12)])
@pytest.mark.parametrize("shape", [(65, 65), (200, 300), (300, 200), (384, 512), (96, 1000)])
def test_svd_block_path(hip, dtype, rtol_s, otol, shape):
  """min(m, n) > 64: block one-sided Jacobi on the MFMA (tnh_svd_block.hip), both dtypes."""
  rng = np.random.default_rng(shape[0] * 7 + shape[1])
  a = rng.standard_normal(shape).astype(dtype)
  _check_svd(hip, a, rtol_s, otol)
  _check_svd(hip, a, rtol_s, otol, max_singular_values=min(shape) // 4)
  # rank-deficient: a zero row/column and an exactly repeated row
  b = a.copy()
  b[3, :] = 0
  b[:, 5] = 0
  b[10, :] = b[11, :]
  _check_svd(hip, b, 20 * rtol_s, 20 * otol, max_singular_values=min(shape) // 4)


@pytest.mark.parametrize("dtype,otol", [(np.float32, 1e-4), (np.float64, 1e-11)])
@pytest.mark.parametrize("shape", [(384, 512), (600, 300)])
def test_svd_topk_mode(hip, dtype, otol, shape):
  """Truncated calls keeping <= half the spectrum skip the rotation accumulation and recover the other
  side from A (tnh_svd_vectors_topk): same factors as the accumulating path, and the automatic
  fallback when the kept values are not all leading ones."""
  rng = np.random.default_rng(shape[0])
  a = rng.standard_normal(shape).astype(dtype)
  k = 40
  d = dev(hip, a)
  u, s, vh, rest = [np.asarray(x) for x in hip.svd(d, 1, max_singular_values=k)]
  hip.svd_topk = False
  try:
    u0, s0, vh0, rest0 = [np.asarray(x) for x in hip.svd(d, 1, max_singular_values=k)]
  finally:
    hip.svd_topk = True
  np.testing.assert_allclose(s, s0, rtol=1e-5 if dtype == np.float32 else 1e-12)
  np.testing.assert_allclose(rest, rest0, rtol=1e-5 if dtype == np.float32 else 1e-12, atol=1e-6)
  np.testing.assert_allclose(u.T @ u, np.eye(k), atol=otol)
  np.testing.assert_allclose(vh @ vh.T, np.eye(k), atol=otol)
  np.testing.assert_allclose((u * s) @ vh, (u0 * s0) @ vh0, atol=otol * np.abs(a).max() * 30)
  # graded spectrum: s_k << s_1 -> the recovered side is re-orthonormalised by the K9 QR (no second factorisation)
  q1, _ = np.linalg.qr(rng.standard_normal((shape[0], shape[0])))
  q2, _ = np.linalg.qr(rng.standard_normal((shape[1], shape[1])))
  r = min(shape)
  spec = 10.0 ** (-np.arange(r) / 8.0)
  g = ((q1[:, :r] * spec) @ q2[:r]).astype(dtype)
  u, s, vh, _ = [np.asarray(x) for x in hip.svd(dev(hip, g), 1, max_singular_values=k)]
  np.testing.assert_allclose(s, spec[:k], rtol=2e-3 if dtype == np.float32 else 1e-9, atol=1e-6 if dtype == np.float32 else 1e-14)
  np.testing.assert_allclose(u.T @ u, np.eye(k), atol=otol * 3)
  np.testing.assert_allclose(vh @ vh.T, np.eye(k), atol=otol * 3)
  best = (q1[:, :k] * spec[:k]) @ q2[:k]                      # best rank-k approximation
  np.testing.assert_allclose((u * s) @ vh, best, atol=otol * 3)   # backward error O(eps * s_1)
  # exactly rank-deficient inside the kept block -> accumulating path with basis completion
  low = (q1[:, :5] * spec[:5]) @ q2[:5]
  u, s, vh, _ = [np.asarray(x) for x in hip.svd(dev(hip, low.astype(dtype)), 1, max_singular_values=k)]
  np.testing.assert_allclose(u.T @ u, np.eye(k), atol=otol * 3)
  np.testing.assert_allclose(vh @ vh.T, np.eye(k), atol=otol * 3)
  np.testing.assert_allclose((u * s) @ vh, low, atol=otol * 3)


def test_svd_prescribed_spectrum_512(hip):
  # SURVEY 8d config-3 input (ii): s_i = 2^(-i/32), Haar factors
  rng = np.random.default_rng(4)
  n = 512
  q1, _ = np.linalg.qr(rng.standard_normal((n, n)))
  q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
  spec = 2.0 ** (-np.arange(n) / 32)
  a = ((q1 * spec) @ q2).astype(np.float32)
  u, s, vh, rest = _check_svd(hip, a, 1e-5, 1e-4, max_singular_values=n // 16)
  np.testing.assert_allclose(s, spec[:n // 16], rtol=1e-4)
  np.testing.assert_allclose(rest, spec[n // 16:], atol=1e-5)


def test_permute_more_than_2p24_tiles(hip):
  """Regression: tiny (a, b) dims under a huge batch used to need > 2^24 workgroups, which the
  runtime rejects (gridDim.x * blockDim.x must stay below 2^32); the kernels now grid-stride."""
  B = (1 << 24) + 3
  x = hip.device_random((16, B, 16), dtype=np.float16, seed=5, normal=True)
  y = hip.transpose(x, (2, 1, 0))
  assert y.shape == (16, B, 16)
  for b in (0, 1, 12345, (1 << 22) + 7, (1 << 24) - 1, (1 << 24), B - 1):
    xi = np.asarray(hip.getitem(x, (slice(None), slice(b, b + 1), slice(None))))
    yi = np.asarray(hip.getitem(y, (slice(None), slice(b, b + 1), slice(None))))
    np.testing.assert_array_equal(yi[:, 0, :], xi[:, 0, :].T)


# ------------------------------------------------------------------ integer dtypes
@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_integer_tensordot_matmul_sum_trace_exact(hip, dtype):
  """The reference hands any NumPy dtype to np.tensordot / np.matmul / np.sum / np.trace
  (numpy_backend.py:35-54, 603-612, 684-707); integer results must be bit-exact, wrap-around included."""
  rng = np.random.default_rng(31)
  a = rng.integers(-9, 10, size=(5, 7, 6, 3)).astype(dtype)
  b = rng.integers(-9, 10, size=(6, 4, 5, 8)).astype(dtype)
  got = hip.tensordot(dev(hip, a), dev(hip, b), [[0, 2], [2, 0]])
  assert got.dtype == np.dtype(dtype)
  np.testing.assert_array_equal(np.asarray(got), np.tensordot(a, b, [[0, 2], [2, 0]]))
  # full contraction -> 0-d, outer product, ragged sizes beyond one 64 x 64 tile
  np.testing.assert_array_equal(np.asarray(hip.tensordot(dev(hip, a), dev(hip, a), [[0, 1, 2, 3], [0, 1, 2, 3]])),
                                np.tensordot(a, a, [[0, 1, 2, 3], [0, 1, 2, 3]]))
  np.testing.assert_array_equal(np.asarray(hip.outer_product(dev(hip, a[0, 0]), dev(hip, b[0, 0]))),
                                np.tensordot(a[0, 0], b[0, 0], 0))
  x = rng.integers(-100, 100, size=(130, 67)).astype(dtype)
  y = rng.integers(-100, 100, size=(67, 71)).astype(dtype)
  np.testing.assert_array_equal(np.asarray(hip.tensordot(dev(hip, x), dev(hip, y), 1)), x @ y)
  # wrap-around like NumPy
  big = np.full((3, 3), np.iinfo(dtype).max // 2 + 5, dtype=dtype)
  with np.errstate(over="ignore"):
    ref = big @ big
  np.testing.assert_array_equal(np.asarray(hip.matmul(dev(hip, big), dev(hip, big))), ref)
  # batched matmul, sum, trace
  p = rng.integers(-9, 10, size=(4, 3, 5)).astype(dtype)
  q = rng.integers(-9, 10, size=(4, 5, 2)).astype(dtype)
  np.testing.assert_array_equal(np.asarray(hip.matmul(dev(hip, p), dev(hip, q))), np.matmul(p, q))
  np.testing.assert_array_equal(np.asarray(hip.sum(dev(hip, a), axis=(1, 3))), a.sum(axis=(1, 3)))
  np.testing.assert_array_equal(np.asarray(hip.sum(dev(hip, a))), a.sum())
  long = rng.integers(-1000, 1000, size=(3, 40000)).astype(dtype)
  np.testing.assert_array_equal(np.asarray(hip.sum(dev(hip, long), axis=1)), long.sum(axis=1))
  t = rng.integers(-9, 10, size=(3, 6, 6)).astype(dtype)
  np.testing.assert_array_equal(np.asarray(hip.trace(dev(hip, t))), np.trace(t, axis1=-2, axis2=-1))


def test_integer_arithmetic_follows_numpy_promotion(hip):
  rng = np.random.default_rng(32)
  a = rng.integers(-50, 50, size=(4, 5)).astype(np.int64)
  b = rng.integers(1, 50, size=(4, 5)).astype(np.int32)
  for f, g in ((hip.addition, np.add), (hip.subtraction, np.subtract), (hip.multiply, np.multiply),
               (hip.divide, np.true_divide)):
    got, ref = f(dev(hip, a), dev(hip, b)), g(a, b)
    assert got.dtype == ref.dtype, (got.dtype, ref.dtype)
    np.testing.assert_allclose(np.asarray(got), ref, rtol=1e-15)
  for got, ref in ((hip.multiply(dev(hip, a), 3), a * 3), (hip.addition(2, dev(hip, a)), 2 + a),
                   (hip.multiply(dev(hip, a), 2.5), a * 2.5), (hip.divide(dev(hip, a), 4), a / 4),
                   (hip.addition(dev(hip, a), dev(hip, a.astype(np.float32))), a + a.astype(np.float32)),
                   (hip.abs(dev(hip, a)), np.abs(a)), (hip.sign(dev(hip, a)), np.sign(a)),
                   (hip.sqrt(dev(hip, np.abs(a))), np.sqrt(np.abs(a))), (hip.conj(dev(hip, a)), a),
                   (hip.ones((2, 3), dtype=np.int64), np.ones((2, 3), dtype=np.int64)),
                   (hip.eye(3, dtype=np.int32), np.eye(3, dtype=np.int32))):
    assert got.dtype == ref.dtype, (got.dtype, ref.dtype)
    np.testing.assert_allclose(np.asarray(got), ref, rtol=1e-15)
  assert float(np.asarray(hip.norm(dev(hip, a)))) == pytest.approx(np.linalg.norm(a))


def test_cast_refuses_to_drop_an_imaginary_part(hip):
  z = dev(hip, np.array([1 + 2j, 3 - 1j]))
  with pytest.raises(TypeError):
    hip.cast(z, np.float64)
  np.testing.assert_array_equal(np.asarray(hip.cast(dev(hip, np.arange(5, dtype=np.int64)), np.float32)),
                                np.arange(5, dtype=np.float32))


def test_device_tensor_deepcopy_pickle_repr(hip):
  import copy, pickle  # pylint: disable=import-outside-toplevel,multiple-imports
  x = np.arange(6.0).reshape(2, 3)
  t = dev(hip, x)
  c = copy.deepcopy(t)
  assert c.ptr != t.ptr
  np.testing.assert_array_equal(np.asarray(c), x)
  np.testing.assert_array_equal(np.asarray(pickle.loads(pickle.dumps(t))), x)
  assert copy.deepcopy(hip) is hip
  assert "[0., 1., 2.]" in repr(t).replace(" ", "").replace(",", ", ").replace(" ", "") or "0." in repr(t)
  assert "data=" not in repr(dev(hip, np.zeros((100,))))


# ------------------------------------------------------------------ transpose-absorbing GEMM (tnh_gemm_view)
def _view_case(hip, dtype, shape_a, shape_b, axes, seed):
  """tensordot through the in-place view kernel vs the permute + NT lowering: bit-identical (same MFMA
  sequence, same K order per tile), and both against float64 on the rounded inputs."""
  rng = np.random.default_rng(seed)
  a = rng.standard_normal(shape_a).astype(np.float32) / 8
  b = rng.standard_normal(shape_b).astype(np.float32) / 8
  if dtype is ta.bfloat16:
    a, b = orc.round_bf16(a), orc.round_bf16(b)
    da, db = hip.to_bfloat16(a), hip.to_bfloat16(b)
  else:
    a, b = a.astype(np.float16), b.astype(np.float16)
    da, db = dev(hip, a), dev(hip, b)
  hip.absorb_transposes = True
  before = (hip.permutes_absorbed, hip.permute_launches)
  got = hip.tensordot(da, db, axes)
  kernel = hip.lib.tnh_gemm_last_kernel().decode()
  absorbed = hip.permutes_absorbed - before[0]
  permutes = hip.permute_launches - before[1]
  hip.absorb_transposes = False
  try:
    ref_dev = hip.tensordot(da, db, axes)
  finally:
    hip.absorb_transposes = True
  g, r = np.asarray(got), np.asarray(ref_dev)
  np.testing.assert_array_equal(g, r)
  ref = np.tensordot(a.astype(np.float64), b.astype(np.float64), axes)
  np.testing.assert_allclose(g, ref, rtol=2.0**-8, atol=2e-3)
  return kernel, absorbed, permutes


@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
def test_gemm_view_absorbs_transposes_bit_exact(hip, dtype):
  # (shape_a, shape_b, axes, expected kernel): >= 192 tiles of 256 x 256, K a multiple of 64
  cases = [
      ((14, 256, 2, 64), (2, 64, 14, 256), ([2, 3], [0, 1]), "bf16_view_nn"),      # config-2 L0: b is [K][N]
      ((14, 2, 256, 64), (64, 14, 2, 256), ([1, 3], [2, 0]), "bf16_view_nn"),      # config-2 L1: two-level rows AND k
      ((2, 64, 14, 256), (2, 64, 14, 256), ([0, 1], [0, 1]), "bf16_view_tt"),      # both k-major
      ((2, 64, 14, 256), (14, 256, 2, 64), ([0, 1], [2, 3]), "bf16_view_tn"),      # a k-major, b K-contiguous
      ((3584, 192), (3584, 192), ([1], [1]), "bf16_view_nt"),                       # plain NT through the view kernel
      ((14, 264, 2, 64), (2, 64, 14, 264), ([2, 3], [0, 1]), "bf16_view_nn"),      # ragged M / N edges (3696 = 14.4 tiles)
      ((2, 64, 15, 248), (2, 64, 15, 248), ([0, 1], [0, 1]), "bf16_view_tt"),      # ragged, k-major clamp on both sides
      ((4, 64, 3600), (3600, 4, 64), ([0, 1], [1, 2]), "bf16_view_tn"),            # rank 3, 4 K-tiles
      # inner contraction runs that are multiples of 32 but not of 64: a K-tile takes its halves from two runs
      ((14, 2, 256, 96), (96, 14, 2, 256), ([1, 3], [2, 0]), "bf16_view_nn"),      # D = 96 flavour of config-2 L1
      ((64, 32, 56, 32), (32, 32, 60, 64), ([1, 3], [0, 1]), "bf16_view_nn"),      # chi = 32: every half its own run
      ((4, 15, 32, 248), (4, 15, 32, 248), ([0, 2], [0, 2]), "bf16_view_tt"),      # both k-major, runs of 32, ragged
      ((12, 2, 304, 160), (160, 12, 2, 304), ([1, 3], [2, 0]), "bf16_view_nn"),    # runs of 160 = 5 halves, ragged
  ]
  for i, (sa, sb, axes, want) in enumerate(cases):
    kernel, absorbed, permutes = _view_case(hip, dtype, sa, sb, axes, 40 + i)
    assert kernel.startswith(want), (sa, sb, axes, kernel)
    assert absorbed == 1 and permutes == 0, (sa, sb, axes, absorbed, permutes)


def test_gemm_view_falls_back_when_it_cannot_read_in_place(hip):
  # inner contraction run of 48 (not a multiple of 32) -> permute + NT, still correct
  # -> both operands permuted to [free, contracted], then the same view kernel on the trivial NT views
  kernel, absorbed, permutes = _view_case(hip, ta.bfloat16, (14, 4, 256, 48), (48, 14, 4, 256), ([1, 3], [2, 0]), 60)
  assert absorbed == 1 and permutes == 2 and kernel.startswith("bf16_view_nt"), (kernel, absorbed, permutes)
  # one side readable in place, the other not: exactly one permute
  kernel, absorbed, permutes = _view_case(hip, ta.bfloat16, (14, 2, 256, 64), (68, 2, 64, 14, 5), ([1, 3], [1, 2]), 62)
  assert absorbed == 1 and permutes == 1, (kernel, absorbed, permutes)
  # a k-major operand above the size policy is permuted instead of being streamed in place
  keep = hip.inplace_max_bytes
  hip.inplace_max_bytes = 1 << 16     # b is k-major and 0.9 MB: above this gate -> permuted, then the plain NT view
  try:
    kernel, absorbed, permutes = _view_case(hip, ta.bfloat16, (14, 256, 2, 64), (2, 64, 14, 256), ([2, 3], [0, 1]), 63)
  finally:
    hip.inplace_max_bytes = keep
  assert absorbed == 1 and permutes == 1 and kernel.startswith("bf16_view_nt"), (kernel, absorbed, permutes)
  # too few tiles for the 256 x 256 kernel
  kernel, absorbed, permutes = _view_case(hip, ta.bfloat16, (512, 128), (128, 512), 1, 61)
  assert absorbed == 0


def test_gemm_view_c_abi_rejects_bad_views_without_launching(hip):
  import ctypes
  a = hip.to_bfloat16(np.zeros((4096, 128), np.float32))
  out = ta.DeviceTensor.empty((4096, 4096), _lib.BF16)
  good = _lib.OperandView(4096, 128, 0, 128, 1, 0)
  for bad in (_lib.OperandView(4096, 128, 0, 96, 1, 0),       # k0 does not divide K
              _lib.OperandView(4096, 128, 0, 16, 1, 128 * 4096),   # k0 not a multiple of 32
              _lib.OperandView(4096, 2, 0, 128, 2, 0),        # no contiguous direction
              _lib.OperandView(4096, 124, 0, 128, 1, 0)):     # rows not 16-byte aligned
    st = hip.lib.tnh_gemm_view(_lib.BF16, _lib.BF16, 4096, 4096, 128, ctypes.c_void_p(a.ptr), ctypes.byref(bad),
                               ctypes.c_void_p(a.ptr), ctypes.byref(good), ctypes.c_void_p(out.ptr), 4096)
    assert st == _lib.ERR_UNSUPPORTED, st
  st = hip.lib.tnh_gemm_view(_lib.F32, _lib.F32, 4096, 4096, 128, ctypes.c_void_p(a.ptr), ctypes.byref(good),
                             ctypes.c_void_p(a.ptr), ctypes.byref(good), ctypes.c_void_p(out.ptr), 4096)
  assert st == _lib.ERR_INVALID


def test_reductions_f32_vector_paths(hip):
  """16-byte-load fast paths of K3/K4: contiguous rows split over many waves, axis sums with a small
  power-of-two inner extent, ragged lengths (scalar tails), against float64."""
  rng = np.random.default_rng(71)
  for shape, axis in [((3, 70001), 1), ((1, 1 << 22), 1), ((5, 4099, 16), 1), ((64, 1000, 4), 1), ((2, 333, 256), 1),
                      ((7, 513, 8), (0, 1)), ((4096, 257), None), ((1031, 64, 32), 1)]:
    x = rng.standard_normal(shape).astype(np.float32)
    got = np.asarray(hip.sum(dev(hip, x), axis=axis))
    ref = x.astype(np.float64).sum(axis=axis)
    scale = np.sqrt(x.size / max(ref.size, 1))
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-6 * scale * 10)
  v = rng.standard_normal(3_000_001).astype(np.float32)
  assert float(np.asarray(hip.norm(dev(hip, v)))) == pytest.approx(float(np.linalg.norm(v.astype(np.float64))), rel=1e-6)


@pytest.mark.parametrize("dtype", [np.bool_, np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16])
def test_narrow_and_unsigned_dtypes_behave_like_numpy(hip, dtype):
  """bool / unsigned / 8-16-bit integers are stored as int64 in HBM with their NumPy dtype as an alias:
  dtype, data movement, modular arithmetic and NumPy's promotion must come out as on the host
  (the reference's tests feed every NumPy dtype to every backend, tests/testing_utils.py:12-20)."""
  rng = np.random.default_rng(5)
  if dtype is np.bool_:
    x = rng.integers(0, 2, size=(3, 4, 5)).astype(dtype)
    y = rng.integers(0, 2, size=(5, 4, 2)).astype(dtype)
  else:
    info = np.iinfo(dtype)
    x = rng.integers(info.min, info.max, size=(3, 4, 5), dtype=dtype, endpoint=True)
    y = rng.integers(info.min, info.max, size=(5, 4, 2), dtype=dtype, endpoint=True)
  dx, dy = dev(hip, x), dev(hip, y)
  assert dx.dtype == np.dtype(dtype)
  np.testing.assert_array_equal(np.asarray(dx), x)
  for got, ref in ((hip.transpose(dx, (2, 0, 1)), np.transpose(x, (2, 0, 1))), (hip.reshape(dx, (12, 5)), x.reshape(12, 5)),
                   (hip.slice(dx, (1, 0, 2), (2, 3, 2)), x[1:3, 0:3, 2:4]), (dx[1], x[1]), (hip.conj(dx), np.conj(x)),
                   (hip.zeros((2, 3), dtype=dtype), np.zeros((2, 3), dtype=dtype)),
                   (hip.ones((2, 3), dtype=dtype), np.ones((2, 3), dtype=dtype)),
                   (hip.eye(3, dtype=dtype), np.eye(3, dtype=dtype)), (hip.diagflat(dx[0, 0]), np.diagflat(x[0, 0]))):
    assert got.dtype == ref.dtype, (got.dtype, ref.dtype)
    np.testing.assert_array_equal(np.asarray(got), ref)
  with np.errstate(over="ignore"):
    refs = [np.tensordot(x, y, [[2, 1], [0, 1]]), np.tensordot(x[0, 0], y[0, 0], 0)]
    gots = [hip.tensordot(dx, dy, [[2, 1], [0, 1]]), hip.outer_product(dx[0, 0], dy[0, 0])]
    if dtype is not np.bool_:
      refs += [x + x, x * x, x - x, x * 3, np.sum(x, axis=1), np.sum(x), np.trace(x[:, :3, :3], axis1=-2, axis2=-1)]
      gots += [hip.addition(dx, dx), hip.multiply(dx, dx), hip.subtraction(dx, dx), hip.multiply(dx, 3),
               hip.sum(dx, axis=1), hip.sum(dx), hip.trace(dev(hip, np.ascontiguousarray(x[:, :3, :3])))]
    if dtype not in (np.bool_, np.uint64):    # uint64 above 2^63 has no exact int64 / float64 image
      refs += [x + x.astype(np.int32), x * np.float32(1.0) * x.astype(np.float32)]
      gots += [hip.addition(dx, dev(hip, x.astype(np.int32))),
               hip.multiply(hip.multiply(dx, 1.0), dev(hip, x.astype(np.float32)))]
  for got, ref in zip(gots, refs):
    if ref.dtype.kind == "f":
      assert got.dtype.kind == "f"
      np.testing.assert_allclose(np.asarray(got), ref, rtol=1e-6)
    else:
      assert got.dtype == ref.dtype, (got.dtype, ref.dtype)
      np.testing.assert_array_equal(np.asarray(got), ref)


def test_gemm_view_tail_split_matches_unsplit(hip):
  """36 x 36 tiles on 256 CUs = 5.06 waves: the last tile row is computed by a split-K launch of the view
  kernel (7 K-slices, f32 partial slabs, fixed-order sum).  Same values as the un-split launch up to the
  rounding of the f32 partial sums, and both right against float64 on sampled entries."""
  rng = np.random.default_rng(77)
  m = n = 9216
  k = 6144
  a = hip.device_random((m, k), dtype=ta.bfloat16, seed=5, normal=True, b=k ** -0.5)
  b = hip.device_random((k, n), dtype=ta.bfloat16, seed=6, normal=True, b=1.0)
  hip.inplace_max_bytes, keep = 1 << 40, hip.inplace_max_bytes
  try:
    got = hip.tensordot(a, b, 1)                      # a K-contiguous, b k-major, tail split on
    assert hip.lib.tnh_gemm_last_kernel().decode() == "bf16_view_nn_256x256x64_pp+tail_splitk"
    _lib.check(hip.lib.tnh_gemm_set_variant(b"auto:t0"))
    try:
      ref_dev = hip.tensordot(a, b, 1)
    finally:
      _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
  finally:
    hip.inplace_max_bytes = keep
  rows = np.concatenate([rng.integers(0, m, 24), np.array([0, m - 256, m - 255, m - 1])])   # incl. the split tail rows
  cols = rng.integers(0, n, 32)
  g = np.stack([np.asarray(hip.getitem(got, (int(r),)))[cols] for r in rows]).astype(np.float64)
  u = np.stack([np.asarray(hip.getitem(ref_dev, (int(r),)))[cols] for r in rows]).astype(np.float64)
  a_rows = np.stack([np.asarray(hip.getitem(a, (int(r),))) for r in rows]).astype(np.float64)
  b_cols = np.stack([np.asarray(hip.getitem(b, (slice(None), int(c)))) for c in cols]).astype(np.float64)
  ref = a_rows @ b_cols.T
  np.testing.assert_allclose(g, ref, rtol=2.0**-8, atol=2.0**-9)
  np.testing.assert_allclose(u, ref, rtol=2.0**-8, atol=2.0**-9)
  np.testing.assert_allclose(g, u, rtol=2.0**-7, atol=1e-6)      # at most one bf16 ulp apart (different f32 summation order)
  assert np.array_equal(g[:24], u[:24]) or np.abs(g[:24] - u[:24]).max() <= 2.0**-7 * np.abs(u[:24]).max()


@pytest.mark.parametrize("ta_,tb_", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(256, 384, 480), (132, 200, 100), (4, 8, 36), (1000, 260, 68), (128, 128, 32)])
def test_gemm_f32_v2_vector_path(hip, ta_, tb_, m, n, k):
  """f32 fast path (16-byte loads, BK = 32, two LDS stages): all four storage forms, ragged M / N edges and a K
  tail, against float64; the kernel actually taken is checked."""
  out, ref, kernel, sk = _gemm_case(hip, np.float32, m, n, k, ta_, tb_, rng=np.random.default_rng(m + n + k + ta_ + 2 * tb_))
  assert kernel == "mfma_f32_128x128x32_v2", kernel
  tol = GEMM_TOL[np.float32]
  np.testing.assert_allclose(out, ref, rtol=tol * sk, atol=tol * sk * np.sqrt(k))


def test_gemm_f32_v2_batched_and_unaligned_fallback(hip):
  rng = np.random.default_rng(91)
  a = rng.standard_normal((3, 140, 64)).astype(np.float32)
  b = rng.standard_normal((3, 64, 72)).astype(np.float32)
  out = np.asarray(hip.matmul(dev(hip, a), dev(hip, b)))
  assert hip.lib.tnh_gemm_last_kernel().decode() == "mfma_f32_128x128x32_v2"
  np.testing.assert_allclose(out, np.matmul(a.astype(np.float64), b.astype(np.float64)), rtol=1e-5, atol=1e-4)
  x = rng.standard_normal((130, 129)).astype(np.float32)          # leading dimension 129: not a multiple of 4
  y = rng.standard_normal((129, 70)).astype(np.float32)
  out = np.asarray(hip.tensordot(dev(hip, x), dev(hip, y), 1))
  assert hip.lib.tnh_gemm_last_kernel().decode() == "mfma_f32_128x128x16"
  np.testing.assert_allclose(out, x.astype(np.float64) @ y.astype(np.float64), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("ta_,tb_", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(1536, 1664, 200), (1410, 1538, 66), (2048, 1024, 16)])
def test_gemm_f64_v2_vector_path(hip, ta_, tb_, m, n, k):
  """f64 fast path (>= 128 tiles of 128 x 128; 16-byte loads, BK = 16, two LDS stages): all four storage forms,
  ragged M / N edges and a K tail, against NumPy's float64 product."""
  out, ref, kernel, sk = _gemm_case(hip, np.float64, m, n, k, ta_, tb_, rng=np.random.default_rng(m + n + k + ta_ + 2 * tb_))
  assert kernel == "mfma_f64_128x128x16_v2", kernel
  tol = GEMM_TOL[np.float64]
  np.testing.assert_allclose(out, ref, rtol=tol * sk * 4, atol=tol * sk * np.sqrt(k) * 4)


@pytest.mark.parametrize("m,n,k,kn", [(4352, 4352, 192, False), (8192, 8192, 128, False), (4400, 5000, 320, False),
                                      (4352, 4608, 256, True), (8192, 4096, 64 * 5, True)])
def test_gemm_persistent_tiles_match_one_workgroup_per_tile(hip, m, n, k, kn):
  """More output tiles than CUs: one workgroup per CU loops over its tiles, the loads of the next tile issued in
  front of the epilogue of the current one (gemm_nt_pp_kernel).  Same MFMA sequence per tile, so the results are
  bit-identical to the one-workgroup-per-tile launch (knob ":g0"), for the NT form (ragged edges included) and
  for the in-place view kernels; and right against float64 on sampled rows."""
  rng = np.random.default_rng(m + n + k)
  a = hip.device_random((m, k), dtype=ta.bfloat16, seed=15, normal=True, b=k ** -0.5)
  b = hip.device_random((k, n) if kn else (n, k), dtype=ta.bfloat16, seed=16, normal=True, b=1.0)
  axes = 1 if kn else [[1], [1]]
  got = hip.tensordot(a, b, axes)
  name = hip.lib.tnh_gemm_last_kernel().decode()
  assert "256x256x64_pp" in name, name
  _lib.check(hip.lib.tnh_gemm_set_variant(b"auto:g0"))
  try:
    ref_dev = hip.tensordot(a, b, axes)
  finally:
    _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
  g, u = np.asarray(got), np.asarray(ref_dev)
  np.testing.assert_array_equal(g, u)
  rows = np.concatenate([rng.integers(0, m, 12), np.array([0, m - 1])])
  an, bn = np.asarray(a).astype(np.float64), np.asarray(b).astype(np.float64)
  ref = an[rows] @ (bn if kn else bn.T)
  np.testing.assert_allclose(g[rows], ref, rtol=2.0**-7, atol=2.0**-8)


def test_gemm_view_tail_split_cuts_as_many_tile_rows_as_it_takes(hip):
  """81 x 7 tiles (the 20736 x 1728 x 20736 product of the D = 12 network) = 2.2 waves on 256 CUs: eight tile rows are
  cut off so that the main part is exactly two waves, and computed by a 4-slice split-K launch.  Same values as the
  un-split launch up to the rounding of the f32 partial sums; both right against float64 on sampled entries."""
  rng = np.random.default_rng(78)
  m, n, k = 20736, 1728, 20736
  a = hip.device_random((m, k), dtype=ta.bfloat16, seed=7, normal=True, b=k ** -0.5)
  b = hip.device_random((n, k), dtype=ta.bfloat16, seed=8, normal=True, b=1.0)
  got = hip.tensordot(a, b, [[1], [1]])
  assert hip.lib.tnh_gemm_last_kernel().decode() == "bf16_view_nt_256x256x64_pp+tail_splitk"
  _lib.check(hip.lib.tnh_gemm_set_variant(b"auto:t0"))
  try:
    ref_dev = hip.tensordot(a, b, [[1], [1]])
    assert hip.lib.tnh_gemm_last_kernel().decode() == "bf16_view_nt_256x256x64_pp"
  finally:
    _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
  rows = np.concatenate([rng.integers(0, m, 16), np.array([0, m - 2048 - 1, m - 2048, m - 1025, m - 1])])   # main / cut rows
  cols = rng.integers(0, n, 32)
  g = np.stack([np.asarray(hip.getitem(got, (int(r),)))[cols] for r in rows]).astype(np.float64)
  u = np.stack([np.asarray(hip.getitem(ref_dev, (int(r),)))[cols] for r in rows]).astype(np.float64)
  a_rows = np.stack([np.asarray(hip.getitem(a, (int(r),))) for r in rows]).astype(np.float64)
  b_rows = np.stack([np.asarray(hip.getitem(b, (int(c),))) for c in cols]).astype(np.float64)
  ref = a_rows @ b_rows.T
  np.testing.assert_allclose(g, ref, rtol=2.0**-8, atol=2.0**-9)
  np.testing.assert_allclose(u, ref, rtol=2.0**-8, atol=2.0**-9)
  np.testing.assert_allclose(g, u, rtol=2.0**-7, atol=1e-6)      # at most one bf16 ulp apart (different f32 summation order)


def test_gemm_view_tail_split_with_two_level_rows_and_runs_of_96(hip):
  """config-2 layout L1 at D = 96: a[i0, k1, i2, k3] . b[k3, j1, k1, j3] -- both operands in place (inner contraction
  runs of 96 = three half K-tiles, two-level rows), 36 x 36 tiles, and the last tile row of the two-level-row operand
  goes through the split-K tail launch (row offset instead of a pointer offset).  Against the un-split launch and
  float64 on sampled entries."""
  rng = np.random.default_rng(78)
  D = 96
  a = hip.device_random((D,) * 4, dtype=ta.bfloat16, seed=25, normal=True, b=1.0 / D)
  b = hip.device_random((D,) * 4, dtype=ta.bfloat16, seed=26, normal=True, b=1.0)
  before = hip.permute_launches
  got = hip.tensordot(a, b, [[1, 3], [2, 0]])
  assert hip.lib.tnh_gemm_last_kernel().decode() == "bf16_view_nn_256x256x64_pp+tail_splitk"
  assert hip.permute_launches == before
  _lib.check(hip.lib.tnh_gemm_set_variant(b"auto:t0"))
  try:
    ref_dev = hip.tensordot(a, b, [[1, 3], [2, 0]])
  finally:
    _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
  an, bn = np.asarray(a).astype(np.float64), np.asarray(b).astype(np.float64)
  i0 = np.concatenate([rng.integers(0, D, 6), np.array([0, D - 3, D - 2, D - 1])])     # incl. the split tail rows
  g = np.asarray(got)[i0][:, -8:]
  u = np.asarray(ref_dev)[i0][:, -8:]
  ref = np.einsum("xkil,ljkm->xijm", an[i0][:, :, -8:, :], bn)
  np.testing.assert_allclose(g, ref, rtol=2.0**-8, atol=2.0**-9)
  np.testing.assert_allclose(u, ref, rtol=2.0**-8, atol=2.0**-9)
  np.testing.assert_allclose(g, u, rtol=2.0**-7, atol=1e-6)


def test_narrow_integer_storage_is_normalised_where_arithmetic_is_not_modular(hip):
  """The int64 storage of a narrow / unsigned / bool tensor may hold an unwrapped value after +, -, * (exact modulo
  2^bits); conversions, sums, traces, division, abs and sign must see NumPy's value (tnh_wrap_int)."""
  a = np.array([200, 100, 255, 7], dtype=np.uint8)
  b = np.array([100, 200, 3, 250], dtype=np.uint8)
  da, db = dev(hip, a), dev(hip, b)
  s = hip.addition(da, db)                                   # stored 300, 300, 258, 257
  with np.errstate(over="ignore"):
    ref = a + b
    np.testing.assert_array_equal(np.asarray(s), ref)
    np.testing.assert_array_equal(np.asarray(hip.cast(s, np.float32)), ref.astype(np.float32))
    np.testing.assert_array_equal(np.asarray(hip.cast(s, np.int64)), ref.astype(np.int64))
    np.testing.assert_array_equal(np.asarray(hip.cast(s, np.int8)), ref.astype(np.int8))
    assert int(np.asarray(hip.sum(s))) == int(np.sum(ref))
    np.testing.assert_allclose(np.asarray(hip.divide(s, db)), ref / b)
    np.testing.assert_allclose(np.asarray(hip.multiply(s, 0.5)), ref * 0.5)
    np.testing.assert_allclose(np.asarray(hip.sqrt(s)), np.sqrt(ref.astype(np.float64)), rtol=1e-12)   # (NumPy itself answers in float16)
    m = hip.multiply(dev(hip, np.full((3, 3), 16, np.uint8)), dev(hip, np.full((3, 3), 17, np.uint8)))   # 272 -> 16
    assert int(np.asarray(hip.trace(m))) == 48
    i8 = hip.multiply(dev(hip, np.array([100, -100, 3], np.int8)), dev(hip, np.array([2, 2, -50], np.int8)))
    r8 = np.array([100, -100, 3], np.int8) * np.array([2, 2, -50], np.int8)        # -56, 56, 106
    np.testing.assert_array_equal(np.asarray(hip.abs(i8)), np.abs(r8))
    np.testing.assert_array_equal(np.asarray(hip.sign(i8)), np.sign(r8))
    np.testing.assert_array_equal(np.asarray(hip.cast(i8, np.float64)), r8.astype(np.float64))
  t = dev(hip, np.array([True, True, False]))
  tt = hip.addition(t, t)                                    # bool + bool is a logical or
  np.testing.assert_array_equal(np.asarray(tt), np.array([True, True, False]))
  assert int(np.asarray(hip.sum(tt))) == 2
  np.testing.assert_array_equal(np.asarray(hip.cast(dev(hip, np.array([0.5, 0.0, -2.0, 1.0])), np.bool_)),
                                np.array([True, False, True, True]))
  np.testing.assert_array_equal(np.asarray(hip.cast(dev(hip, np.array([2, 0, -1], np.int64)), np.bool_)),
                                np.array([True, False, True]))
  from tensornetwork_amd.device_tensor import DeviceTensor
  u = DeviceTensor.from_numpy(np.array([1, 2, 300]), dtype=np.uint8)
  assert u.dtype == np.uint8
  np.testing.assert_array_equal(np.asarray(u), np.array([1, 2, 300]).astype(np.uint8))


def test_row_padded_results_on_the_device(hip):
  """`pad_results` (off by default; measured in round 4 on the MERA chi = 32 layer: 0.2917 s with it, 0.2841 s
  without -- no gain, so it stays off) on the MI355X: a large contraction result whose rows are a power of two is
  written at a padded pitch (tnh_gemm_view's ldc), the next in-place contraction reads it through its strides (row
  index AND contraction index carrying the pitch), everything else sees the dense copy.  Values against NumPy on the
  same bf16-rounded operands."""
  from oracle import numpy_oracle as orc
  rng = np.random.default_rng(12)
  a = orc.round_bf16(rng.standard_normal((13, 256, 128)).astype(np.float32) / 8)      # (r1, r2, k)
  b = orc.round_bf16(rng.standard_normal((128, 4096)).astype(np.float32) / 8)         # (k, c): rows of 8 KiB
  v = orc.round_bf16(rng.standard_normal((4096, 4096)).astype(np.float32) / 64)
  w = orc.round_bf16(rng.standard_normal((12288, 4, 4096)).astype(np.float32) / 64)   # contracts (r2-part, c)
  saved = (hip.pad_results, hip.pad_min_bytes)
  try:
    hip.pad_results, hip.pad_min_bytes = True, 1 << 20
    c1 = hip.tensordot(hip.to_bfloat16(a), hip.to_bfloat16(b), [[2], [0]])             # (13, 256, 4096), padded rows
    assert c1.pad == (2, 4096 + 64) and c1.shape == (13, 256, 4096)
    ref1 = np.tensordot(a.astype(np.float64), b.astype(np.float64), [[2], [0]])
    np.testing.assert_allclose(np.asarray(c1), ref1, rtol=2.0**-7, atol=2.0**-8 * 12)
    c1h = orc.round_bf16(np.asarray(c1)).astype(np.float64)
    before = hip.permute_launches
    c3 = hip.tensordot(c1, hip.to_bfloat16(v), [[2], [0]])                             # rows at the pitch, K contiguous
    assert hip.permute_launches == before and hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_view")
    np.testing.assert_allclose(np.asarray(c3), np.tensordot(c1h, v.astype(np.float64), [[2], [0]]), rtol=2.0**-6,
                               atol=2.0**-7 * 64)
    # the pitch inside the contraction index: (13 x 64) x (4 x 4096) against a dense K-contiguous operand
    c1v = c1.view((13 * 64, 4, 4096))
    before = hip.permute_launches
    c4 = hip.tensordot(hip.to_bfloat16(w), c1v, [[1, 2], [1, 2]])                      # (12288, 832)
    assert hip.permute_launches == before
    ref4 = np.tensordot(w.astype(np.float64), c1h.reshape(13 * 64, 4, 4096), [[1, 2], [1, 2]])
    np.testing.assert_allclose(np.asarray(c4), ref4, rtol=2.0**-6, atol=2.0**-7 * 128)
    # dense consumers
    np.testing.assert_array_equal(np.asarray(hip.transpose(c1, (2, 0, 1))), np.transpose(orc.round_bf16(np.asarray(c1)), (2, 0, 1)))
    np.testing.assert_array_equal(np.asarray(hip.reshape(c1, (13 * 256, 4096))), orc.round_bf16(np.asarray(c1)).reshape(13 * 256, 4096))
    assert hip.reshape(c1, (13 * 256, 4096)).pad == (1, 4160)                          # reshape keeps the padding
  finally:
    hip.pad_results, hip.pad_min_bytes = saved


# ---------------------------------------------------------------- K2 with K <= 16: a store stream
@pytest.mark.parametrize("dtype,m,n,k", [(ta.bfloat16, 1000, 20008, 12), (np.float16, 144, 65536, 4),
                                         (ta.bfloat16, 1728, 24576, 16), (ta.bfloat16, 200, 40000, 8)])
def test_gemm_small_k_store_stream(hip, dtype, m, n, k):
  """One small bond contracted between two tensors (K <= 16, a large result): `gemm_smallk_kernel` -- f32 FMAs in k
  order, so every entry is the correctly rounded half-precision value of the exact product sum (<= 0.5 ulp + the fp32
  round-off), and within one ulp of the tile kernels' result."""
  rng = np.random.default_rng(m + k)
  if hasattr(hip, "_emu"):        # the NumPy emulation of the C ABI (CPU suite): the host path only, at a tenth of the size
    n = n // 80 * 8
  a = rng.standard_normal((m, k)).astype(np.float32)
  b = rng.standard_normal((n, k)).astype(np.float32)
  if dtype is ta.bfloat16:
    a, b = orc.round_bf16(a), orc.round_bf16(b)
    da, db = hip.to_bfloat16(a), hip.to_bfloat16(b)
  else:
    a, b = a.astype(np.float16), b.astype(np.float16)
    da, db = dev(hip, a), dev(hip, b)
  got = np.asarray(hip.tensordot(da, db, [[1], [1]])).astype(np.float64)
  kernel = hip.lib.tnh_gemm_last_kernel().decode()
  if not hasattr(hip, "_emu"):
    assert kernel == "bf16_smallk_64x2048", kernel
  ref = a.astype(np.float64) @ b.astype(np.float64).T
  ulp = 2.0**-7 if dtype is ta.bfloat16 else 2.0**-10            # relative spacing at the bottom of a binade
  assert np.max(np.abs(got - ref) / (np.abs(ref) * ulp + 1e-3 * ulp)) <= 0.52
  if not hasattr(hip, "_emu"):
    _lib.check(hip.lib.tnh_gemm_set_variant(b"bf16_ragged"))
    try:
      tile = np.asarray(hip.tensordot(da, db, [[1], [1]])).astype(np.float64)
      assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_nt_ragged")
    finally:
      _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
    assert np.max(np.abs(got - tile) / (np.abs(ref) * ulp + 1e-3 * ulp)) <= 1.02


# ---------------------------------------------------------------- K2 gather: the long operand read where it lies
# (shape_small, shape_long, axes_small, axes_long, small_first, kernel, same K order as the classic lowering)
_GATHER_CASES = [
    ((12, 12, 12, 12), (12,) * 7, [1, 3], [3, 6], True, "bf16_gather_Sx48", True),            # D = 12: runs of 12 along k
    ((12, 12, 12, 12), (12, 12, 12, 12, 12, 1, 12, 12), [1, 3], [2, 7], True, "bf16_gather_Sx64", True),
    ((12, 12, 12, 12), (12,) * 7, [1, 3], [1, 5], True, "bf16_gather_Sx48", True),            # innermost axis free
    ((12, 12, 12, 12), (12,) * 7, [1, 3], [0, 1], True, "bf16_gather_Sx64", True),            # k-major long operand
    ((12, 12, 12, 12), (12,) * 7, [1, 3], [3, 6], False, "bf16_gather_48xS", True),           # long operand first
    ((12, 12, 12, 12), (12,) * 7, [1, 3], [0, 1], False, "bf16_gather_64xS", True),
    ((16, 8, 8, 8), (8,) * 8, [1, 3], [3, 7], True, "bf16_gather_Sx64", True),                # K = 64, Ms = 128
    ((16, 12, 12, 16), (12, 12, 12, 16, 12, 12), [1, 3], [1, 3], True, "bf16_gather_Sx48", True),   # K = Ms = 192
    ((10, 8, 8, 5), (8, 16, 5, 16, 16, 16), [1, 3], [0, 2], True, "bf16_gather_Sx64", True),  # Ms = 80, K = 40
    ((10, 5, 8, 8), (64, 5, 16, 16, 8), [1, 3], [1, 4], True, "bf16_gather_Sx64", True),      # K = 40, runs of 8
    ((10, 5, 8, 8), (64, 5, 16, 16, 8), [1, 3], [1, 4], False, "bf16_gather_64xS", True),
    ((12, 12, 12, 12), (12,) * 7, [3, 1], [3, 6], True, "bf16_gather_Sx48", False),           # pairs in the other order
]


def _gather_operands(hip, dtype, shape_s, shape_l, seed):
  rng = np.random.default_rng(seed)
  s = rng.standard_normal(shape_s).astype(np.float32) / 8
  l = rng.standard_normal(shape_l).astype(np.float32) / 8
  if dtype is ta.bfloat16:
    s, l = orc.round_bf16(s), orc.round_bf16(l)
    return s, l, hip.to_bfloat16(s), hip.to_bfloat16(l)
  s, l = s.astype(np.float16), l.astype(np.float16)
  return s, l, dev(hip, s), dev(hip, l)


@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
def test_gemm_gather_reads_the_long_operand_in_place_bit_exact(hip, dtype):
  """tensordot of a small tensor with a many-axis one whose contracted axes are not trailing: ONE tnh_gemm_gather
  launch and no K1 pass over the long operand; bit-identical to permute + streaming GEMM when both take the
  contracted pairs in the same order (same MFMA sequence), and against float64 on the rounded inputs."""
  keep = (hip.gather_gemm, hip.gather_min_rows)
  hip.gather_min_rows = 1024
  try:
    for i, (shape_s, shape_l, axes_s, axes_l, small_first, want, same_order) in enumerate(_GATHER_CASES):
      if hasattr(hip, "_emu") and int(np.prod(shape_l)) > (1 << 21):      # the NumPy emulation of the C ABI (CPU suite): two indices of the outermost free axis
        outer = min(ax for ax in range(len(shape_l)) if ax not in axes_l)
        shape_l = tuple(2 if ax == outer else n for ax, n in enumerate(shape_l))
      s, l, ds, dl = _gather_operands(hip, dtype, shape_s, shape_l, 700 + i)
      args = (ds, dl, [axes_s, axes_l]) if small_first else (dl, ds, [axes_l, axes_s])
      hip.gather_gemm = True
      before = (hip.gather_launches, hip.permute_launches)
      got = hip.tensordot(*args)
      kernel = hip.lib.tnh_gemm_last_kernel().decode()
      assert hip.gather_launches - before[0] == 1, (shape_s, shape_l, axes_l, _lib.last_error())
      assert hip.permute_launches - before[1] <= 1          # the small operand at most
      assert kernel == want, (shape_l, axes_l, kernel)
      hip.gather_gemm = False
      classic = hip.tensordot(*args)
      g, c = np.asarray(got), np.asarray(classic)
      ref = np.tensordot(s.astype(np.float64), l.astype(np.float64), [axes_s, axes_l]) if small_first else \
          np.tensordot(l.astype(np.float64), s.astype(np.float64), [axes_l, axes_s])
      assert g.shape == ref.shape
      if same_order:
        np.testing.assert_array_equal(g, c)
      np.testing.assert_allclose(g, ref, rtol=2.0**-8, atol=2e-3)
      np.testing.assert_allclose(c, ref, rtol=2.0**-8, atol=2e-3)
      del got, classic, ds, dl
  finally:
    hip.gather_gemm, hip.gather_min_rows = keep


def test_gemm_gather_follows_the_planner_hint_for_the_small_operand_only(hip):
  """tensordot_planned with the gather lowering: the small operand's free axes take the hinted order (it is permuted
  anyway), the long operand's stay in natural order whatever was hinted, and the returned orders describe the result."""
  keep = (hip.gather_gemm, hip.gather_min_rows)
  hip.gather_gemm, hip.gather_min_rows = True, 4096
  try:
    s, l, ds, dl = _gather_operands(hip, ta.bfloat16, (12, 12, 12, 12), (12,) * 6, 731)
    out, used_s, used_l = hip.tensordot_planned(ds, dl, [[1, 3], [4, 1]], [2, 0], [5, 0, 3, 2])
    assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_gather")
    assert [int(i) for i in used_s] == [2, 0] and [int(i) for i in used_l] == [0, 2, 3, 5]
    ref = np.tensordot(s.astype(np.float64), l.astype(np.float64), [[1, 3], [4, 1]])      # axes (s0, s2, l0, l2, l3, l5)
    np.testing.assert_allclose(np.asarray(out), np.transpose(ref, (1, 0, 2, 3, 4, 5)), rtol=2.0**-8, atol=2e-3)
  finally:
    hip.gather_gemm, hip.gather_min_rows = keep


def test_gemm_gather_may_put_the_long_operands_axes_first(hip):
  """tensordot_planned(..., allow_swap=True): callers for whom the result's axis order is bookkeeping get
  [long operand's axes..., small operand's axes] (the contiguous output tile) and are told so; without the flag the
  order is tensordot's own.  Same values either way."""
  keep = (hip.gather_gemm, hip.gather_min_rows)
  hip.gather_gemm, hip.gather_min_rows = True, 1024
  try:
    s, l, ds, dl = _gather_operands(hip, ta.bfloat16, (12, 12, 12, 12), (12,) * 6, 733)
    axes = [[1, 3], [2, 5]]
    ref = np.tensordot(s.astype(np.float64), l.astype(np.float64), axes)             # axes (s0, s2, l0, l1, l3, l4)
    out, used_s, used_l, swapped = hip.tensordot_planned(ds, dl, axes, None, None, allow_swap=True)
    assert swapped and hip.lib.tnh_gemm_last_kernel().decode() == "bf16_gather_48xS"
    assert [int(i) for i in used_s] == [0, 2] and [int(i) for i in used_l] == [0, 1, 3, 4]
    got = np.asarray(out)
    assert got.shape == (12,) * 6
    np.testing.assert_allclose(got, np.transpose(ref, (2, 3, 4, 5, 0, 1)), rtol=2.0**-8, atol=2e-3)
    plain, _, _, swapped = hip.tensordot_planned(ds, dl, axes, None, None, allow_swap=False) + (False,)
    np.testing.assert_array_equal(np.transpose(np.asarray(plain), (2, 3, 4, 5, 0, 1)), got)
    # the long operand first: nothing to swap
    out, used_l, used_s, swapped = hip.tensordot_planned(dl, ds, [axes[1], axes[0]], None, None, allow_swap=True)
    assert not swapped and hip.lib.tnh_gemm_last_kernel().decode() == "bf16_gather_48xS"
    np.testing.assert_array_equal(np.asarray(out), got)
    # a product outside the gather lowering is never swapped
    out, _, _, swapped = hip.tensordot_planned(ds, dl, [[1, 3], [4, 5]], None, None, allow_swap=True)
    assert not swapped
    np.testing.assert_allclose(np.asarray(out), np.tensordot(s.astype(np.float64), l.astype(np.float64), [[1, 3], [4, 5]]),
                               rtol=2.0**-8, atol=2e-3)
  finally:
    hip.gather_gemm, hip.gather_min_rows = keep


# (shape_small, shape_long, axes_small, axes_long, small_first, kernel): more than 192 contracted indices
_GATHER_KLOOP_CASES = [
    ((12, 12, 12, 12, 12), (12,) * 7, [1, 3, 4], [1, 5, 6], True, "bf16_gather_kloop_Sx64"),     # 144 x 248 832 x 1728
    ((12, 12, 12, 12, 12), (12,) * 7, [1, 3, 4], [1, 5, 6], False, "bf16_gather_kloop_64xS"),
    ((12, 12, 12, 12, 12), (12,) * 7, [1, 3, 4], [0, 4, 5], True, "bf16_gather_kloop_Sx48"),     # innermost axis free
    ((12, 12, 12, 12, 12), (12,) * 7, [1, 3, 4], [0, 4, 5], False, "bf16_gather_kloop_48xS"),
    ((12, 12, 12, 12, 12), (12,) * 7, [1, 3, 4], [2, 5, 6], True, "bf16_gather_kloop_Sx48"),     # the slice's own variant:
    ((12, 12, 12, 12, 12), (12,) * 7, [1, 3, 4], [2, 5, 6], False, "bf16_gather_kloop_48xS"),    # 48 rows, two boxes ahead
    ((16, 6, 8, 64), (6, 16, 8, 64, 16, 4), [1, 3], [0, 3], True, "bf16_gather_kloop_Sx64"),      # Ms = 128, 6 steps of 64
    ((10, 8, 8, 40), (8, 64, 16, 40), [1, 3], [0, 3], True, "bf16_gather_kloop_Sx64"),            # Ms = 80, 8 steps of 40
    ((10, 8, 8, 40), (8, 64, 16, 40), [1, 3], [0, 3], False, "bf16_gather_kloop_64xS"),
]


@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
def test_gemm_gather_k_loop(hip, dtype):
  """More than 192 contracted indices: the box takes the innermost contracted digits, the outermost one is walked step by
  step with the accumulators in registers -- the long operand is still read in place, once.  Against float64 on the
  rounded inputs and against the classic lowering (permute + tile kernels: same products, another grouping of the
  fp32 partial sums, so equal to fp32 round-off, not bit for bit)."""
  keep = (hip.gather_gemm, hip.gather_min_rows)
  hip.gather_min_rows = 1024
  try:
    for i, (shape_s, shape_l, axes_s, axes_l, small_first, want) in enumerate(_GATHER_KLOOP_CASES):
      if hasattr(hip, "_emu") and int(np.prod(shape_l)) > (1 << 21):      # the NumPy emulation of the C ABI (CPU suite)
        outer = min(ax for ax in range(len(shape_l)) if ax not in axes_l)
        shape_l = tuple(2 if ax == outer else n for ax, n in enumerate(shape_l))
      s, l, ds, dl = _gather_operands(hip, dtype, shape_s, shape_l, 760 + i)
      args = (ds, dl, [axes_s, axes_l]) if small_first else (dl, ds, [axes_l, axes_s])
      hip.gather_gemm = True
      before = (hip.gather_launches, hip.permute_launches)
      got = hip.tensordot(*args)
      kernel = hip.lib.tnh_gemm_last_kernel().decode()
      assert hip.gather_launches - before[0] == 1, (shape_s, shape_l, axes_l, _lib.last_error())
      assert hip.permute_launches - before[1] <= 1 and kernel == want, (shape_l, axes_l, kernel)
      hip.gather_gemm = False
      classic = hip.tensordot(*args)
      g, c = np.asarray(got), np.asarray(classic)
      ref = np.tensordot(s.astype(np.float64), l.astype(np.float64), [axes_s, axes_l]) if small_first else \
          np.tensordot(l.astype(np.float64), s.astype(np.float64), [axes_l, axes_s])
      k = int(np.prod([shape_s[a] for a in axes_s]))
      half_ulp = 2.0**-9 if dtype is ta.bfloat16 else 2.0**-12
      tol = dict(rtol=2 * half_ulp, atol=2 * half_ulp * np.abs(ref).max() + 1e-6 * k**0.5)
      np.testing.assert_allclose(g, ref, **tol)
      np.testing.assert_allclose(c, ref, **tol)
      assert np.mean(g == c) > 0.9          # the two fp32 sums round to the same half-precision value almost everywhere
      del got, classic, ds, dl
  finally:
    hip.gather_gemm, hip.gather_min_rows = keep


def test_gemm_gather_leaves_other_products_alone(hip):
  """Outside its range the gather lowering launches nothing: contracted axes already trailing (the streaming kernel
  reads that as it is), an innermost extent that is not a multiple of 4, a short long operand, f32."""
  keep = (hip.gather_gemm, hip.gather_min_rows)
  hip.gather_gemm, hip.gather_min_rows = True, 4096
  try:
    for shape_s, shape_l, axes_s, axes_l in [((12, 12, 12, 12), (12,) * 6, [1, 3], [4, 5]),
                                            ((10, 10, 10, 10), (10,) * 6, [1, 3], [1, 4]),
                                            ((12, 12, 12, 12), (12,) * 4, [1, 3], [0, 2])]:
      s, l, ds, dl = _gather_operands(hip, ta.bfloat16, shape_s, shape_l, 741)
      before = hip.gather_launches
      got = np.asarray(hip.tensordot(ds, dl, [axes_s, axes_l]))
      assert hip.gather_launches == before
      ref = np.tensordot(s.astype(np.float64), l.astype(np.float64), [axes_s, axes_l])
      np.testing.assert_allclose(got, ref, rtol=2.0**-8, atol=2e-3)
    rng = np.random.default_rng(5)
    s32, l32 = rng.standard_normal((12,) * 4).astype(np.float32), rng.standard_normal((12,) * 6).astype(np.float32)
    before = hip.gather_launches
    got = np.asarray(hip.tensordot(dev(hip, s32), dev(hip, l32), [[1, 3], [1, 4]]))
    assert hip.gather_launches == before
    np.testing.assert_allclose(got, np.tensordot(s32.astype(np.float64), l32.astype(np.float64), [[1, 3], [1, 4]]),
                               rtol=1e-4, atol=1e-4)
  finally:
    hip.gather_gemm, hip.gather_min_rows = keep


def test_gemm_gather_c_abi_rejects_bad_descriptors_without_launching(hip):
  import ctypes
  from tensornetwork_amd import hip_backend
  long_shape = (12,) * 5
  desc, bn, rows = hip_backend._gather_descriptor(long_shape, [1, 4])      # pylint: disable=protected-access
  assert (bn, rows) == (48, 1728)
  l = hip.to_bfloat16(np.zeros(long_shape, np.float32))
  s = hip.to_bfloat16(np.zeros((144, 144), np.float32))
  out = ta.DeviceTensor.empty((144, 1728), _lib.BF16)

  def call(d, code=_lib.BF16, k=144, nl=1728, l_elems=12**5):
    return hip.lib.tnh_gemm_gather(code, 144, k, nl, ctypes.c_void_p(s.ptr), 144, ctypes.c_void_p(l.ptr), l_elems,
                                   ctypes.byref(d), ctypes.c_void_p(out.ptr), nl, 1)

  def variant(**changes):
    d = _lib.GatherDesc.from_buffer_copy(bytes(desc))
    for name, (index, value) in changes.items():
      if index is None:
        setattr(d, name, value)
      else:
        getattr(d, name)[index] = value
    return d

  assert call(desc) == _lib.OK
  assert call(variant(mult=(1, 2))) == _lib.ERR_UNSUPPORTED           # two chunks on one image element
  assert call(variant(stride=(1, 10))) == _lib.ERR_UNSUPPORTED        # a piece that is not 8-byte aligned
  assert call(variant(ext=(0, 8))) == _lib.ERR_UNSUPPORTED            # innermost extent: contracted columns != K
  assert call(variant(nd=(None, 9))) == _lib.ERR_UNSUPPORTED
  assert call(variant(text=(0, 5))) == _lib.ERR_UNSUPPORTED           # tiles x BN != long rows
  assert call(desc, l_elems=12**5 - 1) == _lib.ERR_UNSUPPORTED        # the last box leaves the tensor
  assert call(desc, k=136) == _lib.ERR_UNSUPPORTED
  assert call(desc, code=_lib.F32) == _lib.ERR_INVALID


@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
def test_gemm_view_k_walk_forms_are_bit_identical(hip, dtype):
  """Round 5: the view kernel walks the contraction index in the cheapest form both operands allow (KW in
  tnh_gemm_bf16.hip: 2 = one contiguous run, 1 = runs that are multiples of 64, 0 = half K-tiles) -- measured +5 % on
  the headline shape.  Same MFMA sequence in all of them: for operands that allow form 2 / 1, capping the form with
  the A/B knob ":w<d>" must give bit-identical results, and all of them the permute + NT result."""
  rng = np.random.default_rng(31)
  cases = [
      # (shape_a, shape_b, axes): K-contiguous on both sides
      ((3584, 4, 64), (3840, 4, 64), ([1, 2], [1, 2])),              # one contiguous run of 256 each -> form 2
      ((14, 4, 256, 128), (15, 4, 256, 128), ([1, 3], [1, 3])),      # runs of 128 (two-level K and rows) -> form 1
      ((14, 4, 256, 96), (15, 4, 256, 96), ([1, 3], [1, 3])),        # runs of 96 = 3 halves -> form 0 only
      ((14, 12, 256, 32), (15, 12, 256, 32), ([1, 3], [1, 3])),      # runs of 32: every K-tile takes its halves from two runs
      ((3700, 320), (3600, 320), ([1], [1])),                        # ragged M / N edges, odd K-tile count
      # k-major operands (stored [k][row]): the lean loop stages each half of a K-tile from its own scalar base
      ((14, 256, 2, 64), (2, 64, 15, 256), ([2, 3], [0, 1])),        # b k-major (config-2 L0)
      # round 6: b k-major with contraction runs that are multiples of 64 -> the interleaved whole-K-tile lean loop
      # ("auto"; ":w0" caps it to the half-K-tile k-major loop of round 5)
      ((14, 256, 4, 128), (4, 15, 128, 256), ([2, 3], [0, 2])),      # b: two runs levels (4 x 128), two-level rows
      ((3700, 320), (320, 3608), ([1], [0])),                        # ragged M / N edges (N % 8 == 0), 5 K-tiles
      ((15, 248, 3, 64), (3, 64, 14, 264), ([2, 3], [0, 1])),        # ragged edges inside two-level rows, 3 K-tiles
      ((2, 64, 14, 256), (15, 256, 2, 64), ([0, 1], [2, 3])),        # a k-major
      ((2, 96, 14, 256), (2, 96, 15, 256), ([0, 1], [0, 1])),        # both k-major, runs of 96
      ((4, 32, 15, 248), (4, 32, 15, 248), ([0, 1], [0, 1])),        # both k-major, ragged edges, runs of 32
  ]
  for shape_a, shape_b, axes in cases:
    a = (rng.standard_normal(shape_a) / 8).astype(np.float32)
    b = (rng.standard_normal(shape_b) / 8).astype(np.float32)
    if dtype is ta.bfloat16:
      a, b = orc.round_bf16(a), orc.round_bf16(b)
      da, db = hip.to_bfloat16(a), hip.to_bfloat16(b)
    else:
      a, b = a.astype(np.float16), b.astype(np.float16)
      da, db = dev(hip, a), dev(hip, b)
    outs = {}
    for knob in ("auto", "auto:l0", "auto:l2", "auto:w1", "auto:w0", "auto:w0:l0"):
      _lib.check(hip.lib.tnh_gemm_set_variant(knob.encode()))
      try:
        before = hip.permute_launches
        outs[knob] = np.asarray(hip.tensordot(da, db, axes))
        kernel = hip.lib.tnh_gemm_last_kernel().decode()
        assert "view_" in kernel and hip.permute_launches == before, (knob, kernel)
      finally:
        _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
    # ":l0" = the loop of rounds 1-4, ":l2" = lean loop with lane offsets from the operand's base instead of the tile's
    # first row, ":w<d>" = cap on the K-walk form
    for knob in ("auto", "auto:l0", "auto:l2", "auto:w1", "auto:w0"):
      np.testing.assert_array_equal(outs[knob], outs["auto:w0:l0"])
    hip.absorb_transposes = False
    try:
      ref_dev = np.asarray(hip.tensordot(da, db, axes))
    finally:
      hip.absorb_transposes = True
    np.testing.assert_array_equal(outs["auto"], ref_dev)
    ref = np.tensordot(a.astype(np.float64), b.astype(np.float64), axes)
    np.testing.assert_allclose(outs["auto"], ref, rtol=2.0**-8, atol=2e-3)


@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
@pytest.mark.parametrize("m,n,k", [(1000, 900, 320), (2560, 2816, 64), (4096, 4096, 1088), (3584, 3840, 128)])
def test_gemm_lean_main_loop_is_bit_identical(hip, dtype, m, n, k):
  """Round 5: the lean main loop of the ping-pong kernel (SADDR LDS-DMA, untracked fragment reads, K loop unrolled
  by two: ~150 instead of 205 instructions per K-tile and wave) issues the same fragment reads, LDS-DMA pieces and
  MFMAs in the same order as the loop it replaces -- bit-identical results with the knob ':l0' (old loop), for ragged
  M / N edges, odd and even K-tile counts, one K-tile, persistent and one-tile-per-workgroup grids; and against
  float64."""
  rng = np.random.default_rng(m + n + k)
  out = {}
  for knob in ("bf16_256pp", "bf16_256pp:l0", "bf16_256pp:g0", "bf16_256pp:l0:g0"):
    o, ref, kernel, _ = _gemm_case(hip, dtype, m, n, k, 0, 1, variant=knob, rng=np.random.default_rng(m + n + k))
    assert kernel == "bf16_nt_256x256x64_pp"
    out[knob] = o
  np.testing.assert_array_equal(out["bf16_256pp"], out["bf16_256pp:l0"])
  np.testing.assert_array_equal(out["bf16_256pp:g0"], out["bf16_256pp:l0"])
  np.testing.assert_array_equal(out["bf16_256pp:l0:g0"], out["bf16_256pp:l0"])
  assert_gemm(out["bf16_256pp"], ref, dtype, k)


@pytest.mark.parametrize("m,n,k", [(8192, 8192, 128), (8232, 8168, 192), (12288, 4096, 1088)])
def test_gemm_next_tile_under_the_draining_stores_is_bit_identical(hip, m, n, k):
  """Round 6: a persistent workgroup of the ping-pong kernel waits at the top of a tile for the prologue's loads only
  (vmcnt(16) after the 16-byte half epilogue, vmcnt(32) after a full f32 tile, everything after a ragged tile) and runs
  its first cluster while the finished tile's stores drain.  Only a wait moved: with the knob ':e0' (the full wait)
  the results must be bit-identical -- several tiles per CU, full and ragged tiles mixed, half and f32 output, plain and
  view entry points, repeated (a race would not show every time)."""
  import ctypes
  from tensornetwork_amd.device_tensor import DeviceTensor
  A = hip.device_random((m, k), dtype=ta.bfloat16, seed=11, normal=True, a=0.0, b=1.0)
  B = hip.device_random((n, k), dtype=ta.bfloat16, seed=12, normal=True, a=0.0, b=1.0)

  def run(knob, view, out_dt):
    C = DeviceTensor.empty((m, n), out_dt)
    _lib.check(hip.lib.tnh_gemm_set_variant(knob.encode()))
    try:
      if view:
        va, vb = _lib.OperandView(m, k, 0, k, 1, 0), _lib.OperandView(n, k, 0, k, 1, 0)
        _lib.check(hip.lib.tnh_gemm_view(_lib.BF16, out_dt, m, n, k, ctypes.c_void_p(A.ptr), ctypes.byref(va),
                                         ctypes.c_void_p(B.ptr), ctypes.byref(vb), ctypes.c_void_p(C.ptr), n))
      else:
        _lib.check(hip.lib.tnh_gemm(_lib.BF16, out_dt, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), k, ctypes.c_void_p(B.ptr), k,
                                    ctypes.c_void_p(C.ptr), n, 1, 0, 0, 0))
      hip.synchronize()
      assert "256x256x64_pp" in hip.lib.tnh_gemm_last_kernel().decode()
    finally:
      _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
    return np.asarray(C)

  for view in (False, True):
    for out_dt in (_lib.BF16, _lib.F32):
      ref = run("auto:e0", view, out_dt)
      for _ in range(3):
        np.testing.assert_array_equal(run("auto", view, out_dt), ref)
  # and the values themselves: sampled rows against float64
  a64, b64 = np.asarray(A).astype(np.float64), np.asarray(B).astype(np.float64)
  got = run("auto", False, _lib.F32)
  rows = np.array([0, 255, 256, m // 2 + 3, m - 1])
  np.testing.assert_allclose(got[rows], a64[rows] @ b64.T, rtol=1e-4, atol=1e-3 * np.sqrt(k))


@pytest.mark.parametrize("m,n,k", [(9400, 1212, 128), (5000, 5200, 192), (2304, 14080, 64), (16640, 256, 256)])
def test_gemm_tile_orders_cover_every_tile_once_and_agree(hip, m, n, k):
  """Round 6: `pick_raster` chooses the tile order of the ping-pong kernels by grid shape and K (super-tiles shared by
  the XCDs, or per-XCD M-grouped ranges with groups of 4 / 8 / 16 / 32 tile rows: knobs ':r1', ':r2' / ':r0' / ':r3' /
  ':r4').  An order is a bijection workgroup -> tile and nothing else: on grids whose tile-row count is no multiple of
  any group size (37 x 5, 20 x 21, 9 x 55, 65 x 1 tiles, ragged edges) every order must write every tile -- the output
  starts as NaN -- and give the same bits; the values are checked against float64."""
  import ctypes
  from tensornetwork_amd.device_tensor import DeviceTensor
  A = hip.device_random((m, k), dtype=ta.bfloat16, seed=21, normal=True, a=0.0, b=1.0)
  B = hip.device_random((n, k), dtype=ta.bfloat16, seed=22, normal=True, a=0.0, b=1.0)
  nan = hip.convert_to_tensor(np.full((m, n), np.nan, dtype=np.float32))
  outs = {}
  for knob in ("bf16_256pp", "bf16_256pp:r0", "bf16_256pp:r1", "bf16_256pp:r2", "bf16_256pp:r3", "bf16_256pp:r4"):
    C = DeviceTensor.empty((m, n), _lib.F32)
    _lib.check(hip.lib.tnh_d2d(ctypes.c_void_p(C.ptr), ctypes.c_void_p(nan.ptr), C.nbytes), "tnh_d2d")
    _lib.check(hip.lib.tnh_gemm_set_variant(knob.encode()))
    try:
      _lib.check(hip.lib.tnh_gemm(_lib.BF16, _lib.F32, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), k, ctypes.c_void_p(B.ptr), k,
                                  ctypes.c_void_p(C.ptr), n, 1, 0, 0, 0))
      hip.synchronize()
      assert hip.lib.tnh_gemm_last_kernel().decode() == "bf16_nt_256x256x64_pp"
    finally:
      _lib.check(hip.lib.tnh_gemm_set_variant(b"auto"))
    outs[knob] = np.asarray(C)
    assert not np.isnan(outs[knob]).any(), knob
  for knob, got in outs.items():
    np.testing.assert_array_equal(got, outs["bf16_256pp:r1"], err_msg=knob)
  a64, b64 = np.asarray(A).astype(np.float64), np.asarray(B).astype(np.float64)
  rows = np.array([0, 255, 256, m // 2 + 1, m - 1])
  np.testing.assert_allclose(outs["bf16_256pp"][rows], a64[rows] @ b64.T, rtol=1e-4, atol=1e-3 * np.sqrt(k))


def test_k_major_operands_take_the_faster_lowering(hip):
  """Round 5 (profiles/r05_kmajor_gate_small.jsonl): reading a k-major operand in place through the half-K-tile loop
  costs a product ~10 %; the backend takes ONE K1 pass instead wherever that pass is cheaper.  Round 6
  (profiles/r06_kmajor_lean_loop.md): a k-major `b` whose contraction runs are multiples of 64 is read by the
  interleaved whole-K-tile loop, which costs nothing below ~0.6 GB: config-2 L0 at D = 64 AND D = 96 stay in place
  (`kmajor_tile_walk = False` brings the round-5 choice back: the pass at D = 96); a k-major `b` with runs of 96 still
  takes the pass.  Same values either way (bit-identical kernels)."""
  hip.kmajor_inplace_penalty = type(hip).kmajor_inplace_penalty      # the default policy (the autouse fixture zeroed it)
  rng = np.random.default_rng(5)
  cases = [((64,) * 4, (64,) * 4, [[2, 3], [0, 1]], True, 0, "bf16_view_nn"),
           ((96,) * 4, (96,) * 4, [[2, 3], [0, 1]], True, 0, "bf16_view_nn"),
           ((96,) * 4, (96,) * 4, [[2, 3], [0, 1]], False, 1, "bf16_view_nt"),
           # b k-major with contraction runs of 96 (three half K-tiles): the half-K-tile loop's 10 % against a 28 us pass
           ((9216, 32, 96), (32, 36, 96, 256), [[1, 2], [0, 2]], True, 1, "bf16_view_nt")]
  for sa, sb, axes, tile_walk, want_permutes, want_kernel in cases:
    scale = 1.0 / np.sqrt(np.prod([sa[i] for i in axes[0]]))
    a = orc.round_bf16(rng.standard_normal(sa, dtype=np.float32) * scale)
    b = orc.round_bf16(rng.standard_normal(sb, dtype=np.float32))
    da, db = hip.to_bfloat16(a), hip.to_bfloat16(b)
    hip.kmajor_tile_walk = tile_walk
    try:
      before = hip.permute_launches
      got_dev = hip.tensordot(da, db, axes)
      kernel = hip.lib.tnh_gemm_last_kernel().decode()
      assert hip.permute_launches - before == want_permutes and kernel.startswith(want_kernel), (sa, sb, tile_walk, kernel)
    finally:
      hip.kmajor_tile_walk = True
    got = np.asarray(got_dev)
    hip.kmajor_inplace_penalty = 0.0
    try:
      other = np.asarray(hip.tensordot(da, db, axes))
      assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_view_nn")
    finally:
      hip.kmajor_inplace_penalty = type(hip).kmajor_inplace_penalty
    np.testing.assert_array_equal(got.reshape(-1)[::97], other.reshape(-1)[::97])
    if sa[0] == 64:
      ref = np.tensordot(a.astype(np.float64), b.astype(np.float64), axes)
      np.testing.assert_allclose(got, ref, rtol=2.0**-8, atol=2e-3)


# --------------------------------------------------------------------- BASELINE sizes under -m gpu
@pytest.mark.parametrize("layout", ["L0", "L1"])
def test_config2_full_size_sampled_entries(hip, layout):
  """BASELINE configs[1] at its FULL size: two rank-4 bf16 nodes, D = 256 (8.6 GB each), two shared bonds, through
  contract_between; 1024 sampled entries (the four corners of the output included) against float64 dot products of
  the device operands, 2^-8 |ref| + 2^-10 rms(ref) -- the check bench.py's `verified` carries, here under -m gpu."""
  D = 256
  A = hip.device_random((D,) * 4, dtype=ta.bfloat16, seed=3, normal=True, a=0.0, b=1.0 / D)
  B = hip.device_random((D,) * 4, dtype=ta.bfloat16, seed=4, normal=True, a=0.0, b=1.0 / D)
  a, b = ta.Node(A, backend=hip), ta.Node(B, backend=hip)
  if layout == "L0":
    a[2] ^ b[0]  # pylint: disable=pointless-statement
    a[3] ^ b[1]  # pylint: disable=pointless-statement
  else:
    a[1] ^ b[2]  # pylint: disable=pointless-statement
    a[3] ^ b[0]  # pylint: disable=pointless-statement
  out = ta.contract_between(a, b)
  kernel = hip.lib.tnh_gemm_last_kernel().decode()
  assert "256x256x64_pp" in kernel, kernel
  res = C.verify_pair(hip, A, B, out.tensor, layout, seed=D)
  assert res["entries"] >= 1024 and res["ok"], res


def test_d512_row_full_size_sampled_entries(hip):
  """north_star's D = 512 row at full size: A(64,128,512,512) . B(512,512,128,64) over the two D = 512 bonds
  (GEMM 8192 x 8192 x 262144), sampled entries against float64 as above."""
  A = hip.device_random((64, 128, 512, 512), dtype=ta.bfloat16, seed=11, normal=True, a=0.0, b=1.0 / 512)
  B = hip.device_random((512, 512, 128, 64), dtype=ta.bfloat16, seed=12, normal=True, a=0.0, b=1.0 / 512)
  a, b = ta.Node(A, backend=hip), ta.Node(B, backend=hip)
  a[2] ^ b[0]  # pylint: disable=pointless-statement
  a[3] ^ b[1]  # pylint: disable=pointless-statement
  out = ta.contract_between(a, b)
  res = C.verify_pair(hip, A, B, out.tensor, "L0", seed=512)
  assert res["entries"] >= 1024 and res["ok"], res


def test_hip_backend_accepts_the_reference_signatures(hip):
  """the boundary against the committed snapshot of abstract_backend.py:27-1046 /
  numpy_backend.py -- same parameter names, order and defaults (driver-verifiable: no reference on the box)."""
  assert C.signature_mismatches(type(hip)) == []


def test_high_rank_tensors(hip):
  """More than 16 axes: the coalescing pre-pass and the multi-pass permutation on the GPU, bit for bit."""
  C.run_high_rank_cases(hip)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gemm_tiny_outputs_one_workgroup(hip, dtype):
  """Round 6: at most 4 x 4 results over K = 1024 ... 2^20 (the full contraction that ends an MPS overlap) are ONE
  workgroup's tree reduction instead of a split-K GEMM of 128 x 128 tiles; every storage form, ragged K."""
  rng = np.random.default_rng(23)
  for (m, n, k, ta_, tb_) in [(1, 1, 65536, 0, 1), (1, 1, 65536, 0, 0), (2, 3, 5000, 0, 0), (2, 3, 5000, 1, 1), (4, 4, 1024, 0, 1),
                              (3, 1, 100001, 1, 0), (1, 4, 1 << 20, 0, 1)]:
    out, ref, kernel, _ = _gemm_case(hip, dtype, m, n, k, ta_, tb_, rng=rng)
    assert kernel == "tiny_1wg", (kernel, m, n, k)
    tol = 2e-6 if dtype == np.float32 else 1e-14
    np.testing.assert_allclose(out, ref, rtol=tol * 4, atol=tol * np.sqrt(k) * 4, err_msg=f"{m}x{n}x{k} {ta_}{tb_}")
  out, ref, kernel, _ = _gemm_case(hip, dtype, 5, 2, 4096, 0, 1, rng=rng)      # five rows: not tiny
  assert kernel != "tiny_1wg"


@pytest.mark.parametrize("dtype", [ta.bfloat16, np.float16])
def test_gemm_tiny_outputs_one_workgroup_half(hip, dtype):
  """The same for bf16 / f16 (the closing 1 x 1 x 1728 product of every slice of the sliced D = 12 network ran 49 us
  through the ragged 128 x 128 tile kernel): operands converted on load, f32 sums, one rounding of the result."""
  rng = np.random.default_rng(24)
  kind = "bf16" if dtype is ta.bfloat16 else "f16"
  for (m, n, k, ta_, tb_) in [(1, 1, 1728, 0, 1), (1, 1, 65536, 0, 1), (2, 3, 5000, 0, 1), (4, 4, 1024, 0, 1), (1, 4, 1 << 18, 0, 1),
                              (3, 2, 4096, 1, 0)]:
    out, ref, kernel, _ = _gemm_case(hip, dtype, m, n, k, ta_, tb_, rng=rng)
    assert kernel == "tiny_1wg", (kernel, m, n, k)
    C.assert_half_gemm_close(out, ref, kind, err_msg=f"{m}x{n}x{k} {ta_}{tb_}")


def test_index_update_with_a_tensor_assignee(hip):
  """VERDICT r5 missing 6: `t[mask] = assignee` with a tensor assignee, compacted on the device (tnh_masked_scatter)."""
  C.run_index_update_tensor_cases(hip)


def test_k1_pass_of_a_long_row_operand_writes_the_k_blocked_form(hip):
  """Round 6 (profiles/r06_k_blocked_operands.md): an operand the view GEMM cannot read in place, whose rows would be
  1 MiB long (K = 2^19 in bf16), is permuted into the K-blocked form [outer contracted, free, inner contracted] and
  read through a two-level contraction view.  Same K order per tile as the row-major form: bit-identical results;
  sampled rows against float64 dot products.  Second case: 36 x 36 tiles (the split-K tail launch walks the blocked
  views too) with a `b` that COULD be read in place but has its rows exactly 1 MiB apart -- the cost rule gives that
  view up for the pass."""
  cases = [((8, 1792, 4, 2, 16384), (3600, 8, 4, 16384), 1, "bf16_view_nt_256x256x64_pp"),
           ((8, 4608, 4, 2, 16384), (9216, 8, 4, 16384), 2, "bf16_view_nt_256x256x64_pp+tail_splitk")]
  axes = ([0, 2, 4], [1, 2, 3])
  k = 8 * 4 * 16384
  for sa, sb, want_permutes, want_kernel in cases:
    a = hip.device_random(sa, dtype=ta.bfloat16, seed=3, normal=True, a=0.0, b=k ** -0.5)
    b = hip.device_random(sb, dtype=ta.bfloat16, seed=4, normal=True, a=0.0, b=1.0)
    before = hip.permute_launches
    got = hip.tensordot(a, b, axes)
    assert hip.permute_launches - before == want_permutes, (sa, hip.permute_launches - before)
    assert hip.lib.tnh_gemm_last_kernel().decode() == want_kernel
    hip.k_blocked_permutes = False
    try:
      ref = hip.tensordot(a, b, axes)
    finally:
      hip.k_blocked_permutes = True
    m1, n = sa[1], sb[0]
    assert got.shape == ref.shape == (m1, 2, n)
    rows = [(0, 0), (5, 1), (m1 - 1, 1), (m1 - 130, 0)]
    a_t = hip.transpose(a, (1, 3, 0, 2, 4))             # [m1, 2, K]
    for r0, r1 in rows:
      g = np.asarray(hip.getitem(got, (r0, r1)))
      u = np.asarray(hip.getitem(ref, (r0, r1)))
      if want_permutes == 1:
        np.testing.assert_array_equal(g, u)               # same kernel, same K order per tile
      else:                                               # (the tail rows' f32 partial sums are grouped per K-slice)
        np.testing.assert_allclose(g, u, rtol=2.0**-7, atol=1e-6)
      arow = np.asarray(hip.getitem(a_t, (r0, r1))).astype(np.float64).reshape(-1)
      cols = [0, 1, 255, 256, n - 1]
      bcols = np.stack([np.asarray(hip.getitem(b, (c,))).astype(np.float64).reshape(-1) for c in cols])
      np.testing.assert_allclose(g[cols], bcols @ arow, rtol=2.0**-8, atol=2.0**-9)
    del a, b, got, ref, a_t
