"""HipBackend's HOST logic with real numbers on the CPU: the contraction lowering (transpose + reshape + GEMM,
abstract_backend.py:27-38; spec backends/tensorflow/tensordot2.py:22-250), the in-place view lowering with the
planner's layout hints, and the layout planning of network.contract_between / contractors.contract_path -- driven
through tests/emu_tnh.py, a NumPy emulation of the C-ABI entry points those paths call (test infrastructure; the
kernels themselves are checked by the `-m gpu` suite against the same references)."""
import itertools

import numpy as np
import pytest

import tensornetwork_amd as ta
from tensornetwork_amd import _lib, contractors, workloads
from oracle import numpy_oracle as orc
from oracle.numpy_oracle import OracleBackend

from emu_tnh import emulated_backend


def _dev(be, x, bf16=False):
  return be.to_bfloat16(x) if bf16 else be.convert_to_tensor(x)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, "bf16", np.int64])
def test_tensordot_lowering_random_axes(dtype):
  """tensordot2_test.py:163-207 restated for the lowering: random ranks, dims, axes subsets and scalar `axes` --
  every layout branch (contracted axes leading / trailing / in the middle, both K orders) against np.tensordot."""
  rng = np.random.default_rng(11)
  bf16 = dtype == "bf16"
  with emulated_backend() as be:
    for trial in range(80):
      ra, rb = rng.integers(1, 5), rng.integers(1, 5)
      nc = int(rng.integers(0, min(ra, rb) + 1))
      dims_c = [int(rng.integers(1, 7)) for _ in range(nc)]
      axes_a = list(rng.permutation(ra)[:nc])
      axes_b = list(rng.permutation(rb)[:nc])
      sa = [int(rng.integers(1, 7)) for _ in range(ra)]
      sb = [int(rng.integers(1, 7)) for _ in range(rb)]
      for i, (x, y) in enumerate(zip(axes_a, axes_b)):
        sa[x] = sb[y] = dims_c[i]
      if dtype is np.int64:
        a = rng.integers(-9, 9, size=sa).astype(np.int64)
        b = rng.integers(-9, 9, size=sb).astype(np.int64)
      else:
        a = rng.standard_normal(sa).astype(np.float32 if bf16 else dtype)
        b = rng.standard_normal(sb).astype(np.float32 if bf16 else dtype)
        if bf16:
          a, b = orc.round_bf16(a), orc.round_bf16(b)
      axes = [[int(x) for x in axes_a], [int(y) for y in axes_b]]
      if nc and trial % 7 == 0 and axes_a == list(range(ra - nc, ra)) and axes_b == list(range(nc)):
        axes = nc                                     # the scalar form
      got = np.asarray(be.tensordot(_dev(be, a, bf16), _dev(be, b, bf16), axes))
      ref = np.tensordot(a.astype(np.float64), b.astype(np.float64), axes)
      assert got.shape == ref.shape, (sa, sb, axes)
      if dtype is np.int64:
        np.testing.assert_array_equal(got, ref.astype(np.int64))
      else:
        tol = 2.0**-7 if bf16 else (1e-5 if dtype is np.float32 else 1e-12)
        np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * 8 * max(1, int(np.prod(dims_c))) ** 0.5)


_VIEW_CASES = [
    # (shape_a, shape_b, axes, kernel, K1 launches): the cases of tests/test_gpu_kernels.py::test_gemm_view_*
    ((14, 256, 2, 64), (2, 64, 14, 256), ([2, 3], [0, 1]), "bf16_view_nn", 0),      # config-2 L0: b is [K][N]
    ((14, 2, 256, 64), (64, 14, 2, 256), ([1, 3], [2, 0]), "bf16_view_nn", 0),      # config-2 L1: two-level rows and k
    ((2, 64, 14, 256), (2, 64, 14, 256), ([0, 1], [0, 1]), "bf16_view_tt", 0),      # both k-major
    ((2, 64, 14, 256), (14, 256, 2, 64), ([0, 1], [2, 3]), "bf16_view_tn", 0),
    ((3584, 192), (3584, 192), ([1], [1]), "bf16_view_nt", 0),
    ((14, 2, 256, 96), (96, 14, 2, 256), ([1, 3], [2, 0]), "bf16_view_nn", 0),      # runs of 96: half K-tiles
    ((64, 32, 56, 32), (32, 32, 60, 64), ([1, 3], [0, 1]), "bf16_view_nn", 0),      # runs of 32
    ((14, 4, 256, 48), (48, 14, 4, 256), ([1, 3], [2, 0]), "bf16_view_nt", 2),      # runs of 48: both permuted
    ((14, 2, 256, 64), (68, 2, 64, 14, 5), ([1, 3], [1, 2]), "bf16_view_n", 1),     # one side readable in place
]


@pytest.mark.parametrize("sa,sb,axes,kernel,permutes", _VIEW_CASES)
def test_in_place_view_lowering_matches_numpy(sa, sb, axes, kernel, permutes):
  """bf16 products with >= 192 tiles of 256 x 256: which operands are read in place, which are permuted, and the
  values -- the view the host hands to tnh_gemm_view must address exactly the matrix transpose + reshape would have
  built (include/tnh.h: tnh_operand_view)."""
  rng = np.random.default_rng(sum(sa) + sum(sb))
  a = orc.round_bf16(rng.standard_normal(sa).astype(np.float32))
  b = orc.round_bf16(rng.standard_normal(sb).astype(np.float32))
  with emulated_backend() as be:
    da, db = be.to_bfloat16(a), be.to_bfloat16(b)
    before = be.permute_launches
    got = np.asarray(be.tensordot(da, db, axes))
    assert be.lib.tnh_gemm_last_kernel().decode().startswith(kernel)
    assert be.permute_launches - before == permutes
    be.absorb_transposes = False            # the classic lowering: permute + NT GEMM
    ref_classic = np.asarray(be.tensordot(da, db, axes))
  ref = np.tensordot(a.astype(np.float64), b.astype(np.float64), axes)
  k = int(np.prod([sa[i] for i in axes[0]]))
  np.testing.assert_allclose(got, ref, rtol=2.0**-7, atol=2.0**-8 * k**0.5)
  np.testing.assert_allclose(ref_classic, ref, rtol=2.0**-7, atol=2.0**-8 * k**0.5)


def test_planner_hints_decide_the_free_axis_order_and_small_operands_follow_them():
  """tensordot_planned: an operand that is permuted anyway takes the planner's free-axis order; an operand that could
  be read in place still follows the hint when it is small against the result; a large one keeps
  its natural order.  The returned orders describe the result's axes exactly."""
  rng = np.random.default_rng(5)
  a = orc.round_bf16(rng.standard_normal((16, 64, 16, 256)).astype(np.float32))     # free (0, 2), contracted (1, 3)
  b = orc.round_bf16(rng.standard_normal((64, 256, 4096)).astype(np.float32))       # contracted (0, 1), free (2,)
  with emulated_backend() as be:
    da, db = be.to_bfloat16(a), be.to_bfloat16(b)
    out, used_a, used_b = be.tensordot_planned(da, db, [[1, 3], [0, 1]], [2, 0], [2])
    got = np.asarray(out)
    # a (8 MB) is small against the 2 GB-equivalent ... here: 256 x 4096 result of 2 MB -> a is NOT small: natural order
    assert list(used_b) == [2]
    ref = np.tensordot(a.astype(np.float64), b.astype(np.float64), [[1, 3], [0, 1]])          # axes (a0, a2, b2)
    perm = [[0, 2].index(i) for i in used_a] + [2]
    np.testing.assert_allclose(got, np.transpose(ref, perm), rtol=2.0**-7, atol=2.0**-8 * 128)
    # a small operand against a large result: (64, 8, 64) . (64, 64, 8192): m = 8 x ... below the view kernel's range,
    # so use a shape inside it: a = (512, 64, 4), free (0, 2) hinted as (2, 0)
    a2 = orc.round_bf16(rng.standard_normal((512, 64, 4)).astype(np.float32))
    b2 = orc.round_bf16(rng.standard_normal((64, 32768)).astype(np.float32))
    d2a, d2b = be.to_bfloat16(a2), be.to_bfloat16(b2)
    before = be.permute_launches
    out2, used2a, used2b = be.tensordot_planned(d2a, d2b, [[1], [0]], [2, 0], [1])
    assert [int(i) for i in used2a] == [2, 0] and be.permute_launches - before >= 1      # 8 x a.nbytes <= result bytes
    ref2 = np.transpose(np.tensordot(a2.astype(np.float64), b2.astype(np.float64), [[1], [0]]), (1, 0, 2))
    np.testing.assert_allclose(np.asarray(out2), ref2, rtol=2.0**-7, atol=2.0**-8 * 8)
    # the same operand without a hint is read in place: no K1 launch for it
    before = be.permute_launches
    _, used3a, _ = be.tensordot_planned(d2a, d2b, [[1], [0]], None, None)
    assert [int(i) for i in used3a] == [0, 2]


@pytest.mark.parametrize("contractor", [contractors.greedy, contractors.auto])
def test_planned_contraction_paths_reproduce_the_oracle_backend(contractor):
  """contractors.contract_path turns on layout planning for a backend that offers tensordot_planned (edge -> step
  times, operand order swaps, latest-first free axes: network.contract_between): the result of a closed network must
  not depend on it.  20-node 3-regular network and the 16-site MPS overlap against the oracle backend."""
  ob = OracleBackend()
  ref = float(np.asarray(contractor(workloads.random_regular_network(ob, n=20, D=3, seed=4, dtype=np.float64)).tensor))
  with emulated_backend() as be:
    got = float(np.asarray(contractor(workloads.random_regular_network(be, n=20, D=3, seed=4, dtype=np.float64)).tensor))
  assert abs(got - ref) <= 1e-10 * max(1.0, abs(ref))
  kets = workloads.mps_tensors(8, 2, 6, seed=3, dtype=np.float64)
  ref = float(np.asarray(contractor(workloads.mps_overlap_network(ob, kets)).tensor))
  with emulated_backend() as be:
    got = float(np.asarray(contractor(workloads.mps_overlap_network(be, kets)).tensor))
  assert abs(got - ref) <= 1e-10 * max(1.0, abs(ref))


def test_mera_layer_energy_through_the_planned_lowering():
  """Binary-MERA layer energy (simple_mera.py:53-112) at chi = 4, both placements, contractors.branch -- the path the
  chi = 32 / 64 bench legs take -- on the emulated backend against the oracle backend."""
  ham, rho, iso, dis = workloads.mera_random_tensors(4, seed=9, dtype=np.float64)
  branch = lambda nodes: contractors.branch(nodes, nbranch=2)
  ob = OracleBackend()
  ref = float(np.asarray(workloads.mera_energy(ob, ham, rho, iso, dis, branch)))
  with emulated_backend() as be:
    got = float(np.asarray(workloads.mera_energy(be, *(be.convert_to_tensor(t) for t in (ham, rho, iso, dis)), branch)))
  assert abs(got - ref) <= 1e-10 * max(1.0, abs(ref))


def test_every_free_axis_order_a_hint_can_ask_for():
  """All 6 orders of three free axes on each side: the result carries them in the order the call reports."""
  rng = np.random.default_rng(8)
  a = rng.standard_normal((3, 4, 5, 6)).astype(np.float64)       # contracted axis 2
  b = rng.standard_normal((5, 2, 3, 4)).astype(np.float64)       # contracted axis 0
  ref = np.tensordot(a, b, [[2], [0]])                            # (a0, a1, a3, b1, b2, b3)
  with emulated_backend() as be:
    da, db = be.convert_to_tensor(a), be.convert_to_tensor(b)
    for ha, hb in itertools.product(itertools.permutations([0, 1, 3]), itertools.permutations([1, 2, 3])):
      out, ua, ub = be.tensordot_planned(da, db, [[2], [0]], list(ha), list(hb))
      perm = [[0, 1, 3].index(i) for i in ua] + [3 + [1, 2, 3].index(i) for i in ub]
      np.testing.assert_allclose(np.asarray(out), np.transpose(ref, perm), rtol=1e-12, atol=1e-12)


# ---- the golden workload drivers of the GPU suite (tests/cases.py; outputs generated by the reference itself,
# tests/golden/make_golden.py) on the emulated backend: every HipBackend method they reach runs its host logic here
import cases as C  # noqa: E402  pylint: disable=wrong-import-position


def test_golden_ncon_and_contract_between_on_the_emulated_backend(golden):
  with emulated_backend() as be:
    for case in golden.cases["ncon"]:
      C.assert_close(C.run_ncon(be, golden, case), golden[case["out"]])
    for case in golden.cases["contract_between"]:
      C.assert_close(C.run_contract_between(be, golden, case), golden[case["out"]])


def test_golden_contractors_and_misc_ops_on_the_emulated_backend(golden):
  with emulated_backend() as be:
    for case in golden.cases["contractors"]:
      C.assert_close(C.run_contractor(be, golden, case), golden[case["out"]])
    for case in golden.cases["misc"]:
      for name, val in C.run_misc(be, golden, case).items():
        C.assert_close(val, golden[case[name]])
      x, v = be.convert_to_tensor(golden[case["x"]]), be.convert_to_tensor(golden[case["v"]])
      C.assert_close(be.norm(x), golden[case["norm"]])
      C.assert_close(be.sqrt(be.abs(x)), golden[case["sqrtabs"]])
      C.assert_close(be.subtraction(x, v), golden[case["sub"]])


def test_graph_surgery_and_tensor_api_on_the_emulated_backend():
  with emulated_backend() as be:
    C.check_graph_surgery(be, 1e-5)
    C.check_tensor_api(be, 1e-5)


def test_golden_split_node_and_linalg_drivers_on_the_emulated_backend(golden, golden_linalg):
  """split_node in every truncation mode (decompositions.py:38-57 is applied on the HOST), QR / RQ with the phase
  fix of decompositions.py:91-94, eigh / inv / expm built on the SVD and GEMM entry points -- LAPACK stands in for the
  kernels behind the emulated C ABI, the host code is the product's."""
  with emulated_backend() as be:
    for case in golden.cases["split_node"]:
      left, right, trun, recon = C.run_split(be, golden, case)
      assert list(left.shape) == case["left_shape"] and list(right.shape) == case["right_shape"], case["x"]
      x = golden[case["x"]]
      scale = float(np.max(np.abs(x))) * np.sqrt(x.size) + 1e-30
      C.assert_close(trun, golden[case["trun"]], scale=scale)
      C.assert_close(recon, golden[case["recon"]], scale=scale)
    for case in golden_linalg.cases["qr"]:
      x = golden_linalg[case["x"]]
      m = np.asarray(x).reshape(int(np.prod(x.shape[:case["pivot"]])), -1)
      full_rank = np.linalg.matrix_rank(m) == min(m.shape) or not np.any(m)
      C.check_qr_case(be, golden_linalg, case, tight=bool(full_rank))
    for case in golden_linalg.cases["split_qr"]:
      C.check_split_qr_case(be, golden_linalg, case)
    for case in golden_linalg.cases["linalg"]:
      C.check_linalg_case(be, golden_linalg, case)


@pytest.mark.parametrize("tag", C.MPS_GOLDEN_TAGS)
def test_mps_measurements_on_the_emulated_backend(tag):
  """FiniteMPS measurements (base_mps.py:322-479) against the reference's recorded numbers, host code on the
  emulated C ABI."""
  with emulated_backend() as be:
    C.check_mps_golden_case(be, C.load_mps_golden(), tag, 1e-10)


def test_complex_golden_drivers_on_the_emulated_backend(golden_complex):
  """complex64 / complex128 svd (all truncation modes), eigh / inv / expm vs the reference's outputs."""
  with emulated_backend() as be:
    for case in golden_complex.cases["svd"]:
      C.check_svd_case(be, golden_complex, case)
    for case in golden_complex.cases["linalg"]:
      C.check_linalg_case(be, golden_complex, case)


@pytest.mark.parametrize("tag", C.INFINITE_MPS_GOLDEN_TAGS)
def test_infinite_mps_on_the_emulated_backend(tag):
  """InfiniteMPS.canonicalize: Krylov-Schur eigs on 'device' vectors, eigh, masks + index_update, truncated svd, inv."""
  with emulated_backend() as be:
    C.check_infinite_mps_golden_case(be, C.load_mps_golden(), tag, 1e-11)


# ---- selected tests of the `-m gpu` suite, unchanged, with the emulated backend in place of the `hip` fixture: the
# ones whose subject is HOST behaviour (error paths, Krylov recurrences, DMRG sweeps, dtype aliases, the Tensor API)
def _expand(fn):
  """All parameter sets of a test function's stacked @pytest.mark.parametrize marks."""
  sets = [{}]
  for mark in getattr(fn, "pytestmark", []):
    if mark.name != "parametrize":
      continue
    names, values = mark.args[0], mark.args[1]
    names = [n.strip() for n in names.split(",")] if isinstance(names, str) else list(names)
    new = []
    for base in sets:
      for v in values:
        v = v.values if hasattr(v, "values") else v
        vs = (v,) if len(names) == 1 and not (isinstance(v, tuple) and len(names) > 1) else tuple(v)
        new.append({**base, **dict(zip(names, vs))})
    sets = new
  return sets


def _gpu_suite_selection():
  import test_gpu_kernels as TK  # pylint: disable=import-outside-toplevel
  import test_gpu_linalg as TL   # pylint: disable=import-outside-toplevel
  import test_gpu_mps as TM      # pylint: disable=import-outside-toplevel
  import test_gpu_workloads as TW  # pylint: disable=import-outside-toplevel
  import test_gpu_graph as TG      # pylint: disable=import-outside-toplevel
  return [TG.test_graph_surgery_reference_cases_on_device, TG.test_switch_backend_host_network_into_hbm,
          TW.test_config2_bf16_contract_between_full_check, TW.test_config2_bf16_D64_sampled_entries,
          TW.test_tensordot_linearity_and_identity_large, TW.test_regular_network_sliced_on_gpu, TW.test_mera_layer_on_gpu,
          TW.test_hipgraph_capture_replay, TW.test_sliced_contraction_graph_equals_eager,
          TW.test_sliced_bf16_network_accumulates_partials_in_fp32,
          TL.test_linalg_errors, TL.test_eigsh_lanczos_device_vectors, TL.test_eigsh_and_gmres_device_vectors,
          TL.test_eigs_device_vectors, TL.test_eigsh_complex_hermitian_and_pivot, TL.test_compare_and_index_update,
          TL.test_tensor_and_functional_api_on_device, TL.test_complex_split_node_golden, TL.test_complex_qr_rq_golden,
          TM.test_mps_canonical_form_on_device, TM.test_dmrg_ground_energy_vs_exact_f64, TM.test_dmrg_f32_chain_of_12,
          TM.test_free_fermion_2d_one_site_dmrg_on_device,
          TW.test_config1_readme_ncon, TW.test_split_node_config3_small, TW.test_greedy_mps_chain_and_regular_graph,
          TW.test_json_network_from_reference_into_hbm,
          TK.test_gemm_view_absorbs_transposes_bit_exact, TK.test_gemm_view_falls_back_when_it_cannot_read_in_place,
          TK.test_gemm_gather_reads_the_long_operand_in_place_bit_exact,
          TK.test_gemm_gather_follows_the_planner_hint_for_the_small_operand_only,
          TK.test_gemm_gather_may_put_the_long_operands_axes_first, TK.test_gemm_gather_k_loop,
          TK.test_gemm_small_k_store_stream,
          TK.test_gemm_gather_leaves_other_products_alone, TK.test_gemm_gather_c_abi_rejects_bad_descriptors_without_launching,
          TK.test_tensordot_random_axes_property, TK.test_misc_golden, TK.test_tensordot_golden,
          TK.test_tensordot_errors_and_empty, TK.test_elementwise_math, TK.test_init_functions, TK.test_casts,
          TK.test_integer_tensordot_matmul_sum_trace_exact, TK.test_integer_arithmetic_follows_numpy_promotion,
          TK.test_cast_refuses_to_drop_an_imaginary_part, TK.test_device_tensor_deepcopy_pickle_repr,
          TK.test_slice_diagonal_diagflat_bit_exact, TK.test_permute_golden_bit_exact, TK.test_permute_all_perms_rank4,
          TK.test_narrow_and_unsigned_dtypes_behave_like_numpy,
          TK.test_narrow_integer_storage_is_normalised_where_arithmetic_is_not_modular]


def _selection_ids():
  try:
    return [f.__module__.replace("test_gpu_", "") + "::" + f.__name__ for f in _gpu_suite_selection()]
  except Exception:  # pylint: disable=broad-except
    return []


@pytest.mark.parametrize("index", range(len(_selection_ids())), ids=_selection_ids())
def test_gpu_suite_host_level_tests_on_the_emulated_backend(index, golden, golden_linalg, golden_complex):
  import inspect
  fn = _gpu_suite_selection()[index]
  fixtures = {"golden": golden, "golden_linalg": golden_linalg, "golden_complex": golden_complex}
  wanted = inspect.signature(fn).parameters
  for params in _expand(fn):
    with emulated_backend() as be:
      kwargs = {"hip": be, **{k: v for k, v in fixtures.items() if k in wanted}, **params}
      fn(**{k: v for k, v in kwargs.items() if k in wanted})


def test_bench_svd_leg_runs_on_the_emulated_backend():
  """bench.py's split_node timing leg (svd_case / check_svd_case) end to end on a small matrix: the record it writes
  into the JSON line and the LAPACK check it applies."""
  import bench  # pylint: disable=import-outside-toplevel
  rng = np.random.default_rng(2)
  n, k = 64, 8
  mat_host = rng.standard_normal((n, n)).astype(np.float32)
  s_ref = np.linalg.svd(mat_host.astype(np.float64), compute_uv=False)
  with emulated_backend() as be:
    rec, outputs = bench.svd_case(ta, be, be.convert_to_tensor(mat_host), n, k, "natural")
    check = bench.check_svd_case(mat_host.astype(np.float64), s_ref, n, k, outputs)
  assert rec["n"] == n and rec["k"] == k and len(rec["samples_ms"]) == 3 and rec["seconds"] * 1e3 == min(rec["samples_ms"])
  assert check["ok"], check


def test_bench_main_assembles_its_json_line_on_the_emulated_backend(monkeypatch, capsys, tmp_path):
  """bench.py end to end (headline leg with its value check, the sliced 64-node network, the MERA layer, roofline
  and verified objects) at toy sizes on the emulated backend: the control flow and the JSON contract of the line the
  driver parses -- the numbers themselves mean nothing here."""
  import json  # pylint: disable=import-outside-toplevel
  import sys  # pylint: disable=import-outside-toplevel
  import bench  # pylint: disable=import-outside-toplevel
  with emulated_backend() as be:
    monkeypatch.setattr(ta, "get_hip_backend", lambda: be)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--bond", "16", "--steps", "2", "--warmup", "1", "--svd-n", "0",
                                      "--rr-bond", "3", "--rr-bond-small", "2", "--rr-min-slices", "4", "--mera-chi", "4", "--no-sweep",
                                      "--no-extras", "--no-cpu-baseline"])
    for var in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
      monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))       # the detail file goes next to the (relocated) bench
    bench.main()
  out = capsys.readouterr().out.strip().splitlines()
  line = out[-1]
  assert len(out) == 1 and len(line.encode()) < 4096, len(line)      # the driver reads the LAST line of an 8 KB tail
  head = json.loads(line)
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "verified", "sliced_network", "mera", "detail"):
    assert key in head, key
  assert head["n_gpus"] == 1 and head["steps"] == 2 and head["warmup"] == 1 and head["dtype"] == "bf16"
  assert head["higher_is_better"] is True and head["vs_baseline"] is None and "workload" in head["config"]
  assert set(head["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
  assert set(head["verified"]) == {"all_ok", "checks", "failed"}
  # everything else is in the detail file, written next to bench.py and into gpurun_out/ (which travels back)
  for path in (tmp_path / head["detail"], tmp_path / "gpurun_out" / head["detail"]):
    rec = json.loads(path.read_text())
    assert rec["value"] == pytest.approx(head["value"], rel=1e-5) and rec["roofline"]["kernel"] == head["roofline"]["kernel"]
  assert rec["verified"]["headline_D16_L0"]["ok"] is True and rec["verified"]["mera_chi4_bf16_vs_f32"]["ok"] is True
  assert "sliced_network_bf16_vs_f32" in rec["verified"] and "all_ok" in rec["verified"]   # (the statistical model
  # behind the sliced check needs more than the 4 partials of a D = 2 toy network to hold: only its presence is asserted)


def test_bench_compact_line_of_a_full_size_record_stays_under_the_parser_limit():
  """the driver could not parse a 21 KB line.  The compact line of a REAL full-size record (round 3's
  builder-side run, every leg present) must stay under 4 KB, parse, and carry the contract keys + roofline +
  cpu_baseline; a record in which every secondary leg failed must too."""
  import json  # pylint: disable=import-outside-toplevel
  import os  # pylint: disable=import-outside-toplevel
  import bench  # pylint: disable=import-outside-toplevel
  path = os.path.join(os.path.dirname(__file__), "..", "profiles", "r03_bench_final.json")
  full = json.loads(open(path).read().strip().splitlines()[-1])
  assert len(json.dumps(full)) > 16000
  text = bench.compact_line(full, "bench_detail.json")
  assert len(text.encode()) < bench.COMPACT_LINE_LIMIT and "\n" not in text
  head = json.loads(text)
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
    assert key in head, key
  assert set(head["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
  assert set(head["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
  assert head["verified"] == {"all_ok": True, "checks": 10, "failed": []}
  assert head["bond_sweep"]["D512row"][1] == 1.0 and head["svd"]["seconds"] == pytest.approx(0.04169, rel=1e-3)
  broken = dict(full)
  for key in ("sliced_network", "bond_sweep", "mera", "mera_chi64", "svd", "cpu_baseline"):
    broken[key] = {"error": "RuntimeError: " + "x" * 4000}
  text = bench.compact_line(broken, "bench_detail.json")
  assert len(text.encode()) < bench.COMPACT_LINE_LIMIT and json.loads(text)["svd"]["error"].startswith("RuntimeError")


def test_row_padded_contraction_results_are_read_in_place_or_made_dense():
  """`pad_results` (off by default): a large contraction result whose rows are a power of two gets 128 bytes of
  padding per row (profiles/r03_gemm_epilogue.md section 5: power-of-two pitches alias the HBM channels of the
  contraction that re-views the result).  The next in-place contraction reads it as it lies -- its operand view
  carries the pitch -- and everything else sees the dense copy."""
  rng = np.random.default_rng(12)
  a = orc.round_bf16(rng.standard_normal((13, 256, 128)).astype(np.float32) / 8)      # (r1, r2, k)
  b = orc.round_bf16(rng.standard_normal((128, 4096)).astype(np.float32) / 8)         # (k, c): rows of 8 KiB
  w = orc.round_bf16(rng.standard_normal((256, 4096, 16)).astype(np.float32) / 32)    # contracts (r2, c) of the result
  ref1 = np.tensordot(a.astype(np.float64), b.astype(np.float64), [[2], [0]])         # (13, 256, 4096)
  with emulated_backend() as be:
    be.pad_results, be.pad_min_bytes = True, 1 << 20
    c1 = be.tensordot(be.to_bfloat16(a), be.to_bfloat16(b), [[2], [0]])
    assert c1.pad == (2, 4096 + 64) and c1.shape == (13, 256, 4096)
    assert c1.strides == [256 * 4160, 4160, 1]
    np.testing.assert_allclose(np.asarray(c1), ref1, rtol=2.0**-7, atol=2.0**-8 * 12)
    c1h = orc.round_bf16(np.asarray(c1))
    # a small product over (r2, c) of the result: outside the view kernel's range -> the dense copy, classic lowering
    c2 = be.tensordot(be.to_bfloat16(w), c1, [[0, 1], [1, 2]])                           # (16, 13)
    ref2 = np.tensordot(w.astype(np.float64), c1h.astype(np.float64), [[0, 1], [1, 2]])
    np.testing.assert_allclose(np.asarray(c2), ref2, rtol=2.0**-6, atol=2.0**-7 * 1024)
    # a product inside the view kernel's range reads the padded result in place: c1 (13 x 256 rows at the PITCH) . v
    v = orc.round_bf16(rng.standard_normal((4096, 4096)).astype(np.float32) / 64)
    vd = be.to_bfloat16(v)
    n_before = len(be.lib.calls)
    before = be.permute_launches
    c3 = be.tensordot(c1, vd, [[2], [0]])                                                # m = 3328, n = 4096, K = 4096
    call = [c for c in be.lib.calls[n_before:] if c[0] == "view_gemm"]
    assert len(call) == 1 and call[0][4] == (1, 0, 0) and be.permute_launches == before, (call, be.lib.calls[n_before:])
    ref3 = np.tensordot(c1h.astype(np.float64), v.astype(np.float64), [[2], [0]])
    assert c3.pad == (2, 4160)                                                           # 3328 x 4096 again: padded
    np.testing.assert_allclose(np.asarray(c3), ref3, rtol=2.0**-6, atol=2.0**-7 * 64)
    # dense consumers
    np.testing.assert_allclose(np.asarray(be.sum(c1, axis=(0, 1))), c1h.astype(np.float64).sum(axis=(0, 1)), rtol=2e-2, atol=0.5)
    np.testing.assert_array_equal(np.asarray(be.transpose(c1, (2, 0, 1))), np.transpose(c1h, (2, 0, 1)))
    np.testing.assert_array_equal(np.asarray(c1[3, 5:9]), c1h[3, 5:9])
    np.testing.assert_array_equal(np.asarray(be.reshape(c1, (13 * 256, 4096))), c1h.reshape(13 * 256, 4096))
    import copy  # pylint: disable=import-outside-toplevel
    d = copy.deepcopy(c1)
    assert d.pad is None
    np.testing.assert_array_equal(np.asarray(d), c1h)
    with pytest.raises(ValueError):
      c1.view((13 * 256 * 2, 2048))                 # does not keep the row boundary
    assert c1.view((13 * 256, 4096)).pad == (1, 4160) and c1.view((13, 16, 16, 4096)).pad == (3, 4160)
    # off by default
    be.pad_results = False
    assert be.tensordot(be.to_bfloat16(a), be.to_bfloat16(b), [[2], [0]]).pad is None


def test_row_padded_operand_with_the_pitch_in_its_contraction_index():
  """The shape of the MERA layer's second large product: the padded result (R, r2, c) is contracted over (r2, c) --
  the contraction index has two levels and the OUTER one strides by the pitch -- against a dense K-contiguous
  operand.  One view GEMM, no permute, values as NumPy's."""
  from tensornetwork_amd.device_tensor import DeviceTensor  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import _lib  # pylint: disable=import-outside-toplevel
  rng = np.random.default_rng(13)
  R, r2, c, pitch, m = 1024, 4, 4096, 4096 + 64, 12288          # 48 x 4 tiles of 256 x 256, K = 16384
  t = rng.integers(-4, 5, size=(R, r2, c)).astype(np.float32) / 8           # small dyadic values: exact in bf16
  a = rng.integers(-4, 5, size=(m, r2, c), dtype=np.int8).astype(np.float32) / 8
  with emulated_backend() as be:
    padded_host = np.zeros((R * r2, pitch), dtype=np.float32)
    padded_host[:, :c] = t.reshape(R * r2, c)
    block = be.to_bfloat16(padded_host)
    tp = DeviceTensor(block._block, (R, r2, c), _lib.BF16, 0, None, (2, pitch))   # pylint: disable=protected-access
    np.testing.assert_array_equal(np.asarray(tp), t)
    before, n_before = be.permute_launches, len(be.lib.calls)
    out = be.tensordot(be.to_bfloat16(a), tp, [[1, 2], [1, 2]])                    # (m, R)
    calls = [x for x in be.lib.calls[n_before:] if x[0] == "view_gemm"]
    assert be.permute_launches == before and len(calls) == 1
    assert calls[0][5] == (1, 0, pitch), calls                                      # b: sk0 = 1, one row level, sk1 = pitch
    got = np.asarray(out)
  ref = a.reshape(m, -1) @ t.reshape(R, -1).T                                       # f32, exact for these values up to 2^24
  np.testing.assert_allclose(got, ref, rtol=2.0**-7, atol=2.0**-7)


# ---- the band SVD's host logic (hip_backend._svd_band*): which call shapes take the path, the truncation rule on the
#      host between the two device calls, padding to the 16-wide panels, the loud fall-back -- on the emulated C ABI
#      (tests/emu_tnh.py keeps the CONTRACT of tnh_svd_band_*; the kernels are tested on the MI355X)
def _ref_svd_rule(a, max_sv, max_err, relative):
  """decompositions.py:21-74 on a matrix, through the pinned oracle."""
  return orc.svd(a.astype(np.float64), 1, max_sv, max_err, relative)


def _check_band_result(a, out, ref, tag):
  u, s, vh, rest = (np.asarray(x, dtype=np.float64) for x in out)
  ur, sr, vhr, restr = ref
  assert s.shape == sr.shape and rest.shape == restr.shape, (tag, s.shape, sr.shape, rest.shape, restr.shape)
  s0 = max(float(sr[0]) if sr.size else float(restr[0]), 1e-30)
  np.testing.assert_allclose(s, sr, atol=2e-6 * s0, err_msg=tag)
  np.testing.assert_allclose(rest, restr, atol=4e-6 * s0, err_msg=tag)
  k = s.shape[0]
  assert u.shape == (a.shape[0], k) and vh.shape == (k, a.shape[1])
  np.testing.assert_allclose(u.T @ u, np.eye(k), atol=1e-5)
  np.testing.assert_allclose(vh @ vh.T, np.eye(k), atol=1e-5)
  np.testing.assert_allclose((u * s) @ vh, (ur * sr) @ vhr, atol=2e-5 * s0, err_msg=tag)


@pytest.mark.parametrize("m,n", [(512, 512), (640, 512), (512, 768), (520, 520), (515, 700), (700, 523)])
@pytest.mark.parametrize("mode", ["max_sv", "err_only", "both", "full", "relative_err"])
def test_band_svd_host_logic_takes_every_call_shape_the_reference_makes(m, n, mode):
  rng = np.random.default_rng(m * 7 + n)
  r = min(m, n)
  qu, _ = np.linalg.qr(rng.standard_normal((m, r)))
  qv, _ = np.linalg.qr(rng.standard_normal((n, r)))
  spec = 3.0 * 2.0 ** (-np.arange(r) / 40.0)                      # 3 ... 4e-4: every value above the 1e-6 s_1 floor
  a = ((qu * spec) @ qv.T).astype(np.float32)
  kw = {"max_sv": dict(max_singular_values=37), "err_only": dict(max_truncation_error=0.05),
        "both": dict(max_singular_values=300, max_truncation_error=0.3),
        "full": {}, "relative_err": dict(max_truncation_error=1e-2, relative=True)}[mode]
  ref = _ref_svd_rule(a, kw.get("max_singular_values"), kw.get("max_truncation_error"), kw.get("relative", False))
  with emulated_backend() as be:
    be.lib.band_svd = True
    out = [np.asarray(x) for x in be.svd(be.convert_to_tensor(a), 1, **kw)]
    assert be.last_svd_path == "band", (be.last_svd_path, be.last_svd_band_status)
    calls = [c for c in be.lib.calls if c[0].startswith("svd_band")]
    assert [c[0] for c in calls] == ["svd_band_factor", "svd_band_vectors"]
    mm, nn, kcap = calls[0][1]
    assert mm >= nn and nn % 16 == 0 and nn - r == -r % 16 and calls[1][1][:3] == (mm, nn, kcap)
    if "max_truncation_error" in kw:
      assert kcap == nn                           # k is only known after the values
    elif mode == "max_sv":
      assert kcap == (37 + (-r % 16) + 3) // 4 * 4
  _check_band_result(a, out, ref, f"{m}x{n} {mode}")


def test_band_svd_host_logic_falls_back_loudly():
  rng = np.random.default_rng(5)
  low = (rng.standard_normal((512, 6)) @ rng.standard_normal((6, 512))).astype(np.float32)       # rank 6
  graded = ((np.linalg.qr(rng.standard_normal((512, 512)))[0] * 2.0 ** (-np.arange(512) / 8.0))
            @ np.linalg.qr(rng.standard_normal((512, 512)))[0]).astype(np.float32)               # 1 ... 6e-20
  with emulated_backend() as be:
    be.lib.band_svd = True
    be._svd_band_failed = set()      # pylint: disable=protected-access
    # (1) numerically rank-deficient: reported by the factor stage.  The first call only learns it after the vectors
    # stage (nothing is read back in between); the shape is remembered and the second call stops after the factor.
    for trip, n_calls in ((0, 2), (1, 1)):
      be.lib.calls.clear()
      u, s, vh, rest = be.svd(be.convert_to_tensor(low), 1, max_singular_values=8)
      assert be.last_svd_path.startswith("jacobi") and be.last_svd_band_status & 1, trip
      assert len([c for c in be.lib.calls if c[0].startswith("svd_band")]) == n_calls, (trip, be.lib.calls)
      sr = np.linalg.svd(low.astype(np.float64), compute_uv=False)
      np.testing.assert_allclose(np.concatenate([np.asarray(s), np.asarray(rest)]), sr, atol=1e-5 * sr[0])
    # a shape that keeps reporting backs off: after 2 reports the next min(2^2, 64) = 4 calls skip the band path
    # altogether (two-site DMRG splits keep values below the floor on every call), then it is tried again
    assert be._svd_band_backoff[(_lib.F32, 512, 512)] == (2, 4)      # pylint: disable=protected-access
    for _ in range(4):
      be.lib.calls.clear()
      be.svd(be.convert_to_tensor(low), 1, max_singular_values=8)
      assert be.last_svd_path.startswith("jacobi") and not [c for c in be.lib.calls if c[0].startswith("svd_band")]
    be.lib.calls.clear()
    be.svd(be.convert_to_tensor(low), 1, max_singular_values=8)
    assert len([c for c in be.lib.calls if c[0].startswith("svd_band")]) == 1      # tried again (and reported again)
    # (2) a kept value below 1e-6 s_1 (here: a full SVD of a steeply graded matrix): status 16 -> Jacobi
    be._svd_band_failed = set()      # pylint: disable=protected-access
    be._svd_band_backoff = {}        # pylint: disable=protected-access
    u, s, vh, rest = be.svd(be.convert_to_tensor(graded), 1)
    assert be.last_svd_path.startswith("jacobi") and s.shape == (512,) and rest.shape == (0,)
    # ... while the same matrix truncated above the floor stays on the band path
    be._svd_band_backoff = {}        # pylint: disable=protected-access
    u, s, vh, rest = be.svd(be.convert_to_tensor(graded), 1, max_singular_values=64)
    assert be.last_svd_path == "band" and s.shape == (64,) and rest.shape == (448,)
    # (3) too small for the path, and a work buffer beyond the cap
    be.svd(be.convert_to_tensor(graded[:300, :300]), 1, max_singular_values=8)
    assert be.last_svd_path == "jacobi"
    be.svd_band_max_factor_bytes = 512 * 64 * 128
    be.svd(be.convert_to_tensor(graded), 1, max_singular_values=65)
    assert be.last_svd_path == "jacobi"
    # k unknown before the values AND k = n too large for the buffer: values first with the smallest layout, then again
    be.lib.calls.clear()
    u, s, vh, rest = be.svd(be.convert_to_tensor(graded), 1, max_truncation_error=0.1)
    assert be.last_svd_path == "band"
    assert [c[0] for c in be.lib.calls if c[0].startswith("svd_band")] == ["svd_band_factor", "svd_band_vectors"] * 2
    ref = _ref_svd_rule(graded, None, 0.1, False)
    assert s.shape == ref[1].shape and 8 < s.shape[0] <= 64


def test_band_svd_kept_values_come_from_the_refined_brackets():
  """tnh_svd_band_factor's S carries 20 bits per value (5e-7 s_1); the kept values must come from
  the vectors stage's refined brackets -- small kept values keep their RELATIVE accuracy."""
  rng = np.random.default_rng(9)
  q1, _ = np.linalg.qr(rng.standard_normal((512, 512)))
  q2, _ = np.linalg.qr(rng.standard_normal((512, 512)))
  spec = np.concatenate([2.0 ** (-np.arange(64) / 4.0), np.full(448, 2.0 ** -17)])      # kept: 1 ... 1.8e-5
  a = ((q1 * spec) @ q2.T).astype(np.float32)
  with emulated_backend() as be:
    be.lib.band_svd = True
    u, s, vh, rest = [np.asarray(x) for x in be.svd(be.convert_to_tensor(a), 1, max_singular_values=64)]
    assert be.last_svd_path == "band"
  sr = np.linalg.svd(a.astype(np.float64), compute_uv=False)
  np.testing.assert_allclose(np.asarray(s), sr[:64], rtol=2e-6)          # not 5e-7 / 1.8e-5 = 3 %
  np.testing.assert_allclose(np.asarray(rest), sr[64:], atol=1e-6)


@pytest.mark.parametrize("dtype", [np.float64, np.complex64, np.complex128])
@pytest.mark.parametrize("mode", ["max_sv", "err_only", "full"])
def test_band_svd_host_logic_in_the_other_dtypes(dtype, mode):
  """float64 runs the same host logic on the f64 entry points (round 4); complex64 / complex128 go through the real
  embedding (2m x 2n, every value doubled) of the f32 / f64 band path."""
  rng = np.random.default_rng(17)
  m, n = (520, 530) if dtype == np.float64 else (280, 264)
  r = min(m, n)
  cplx = np.dtype(dtype).kind == "c"
  g = lambda *sh: rng.standard_normal(sh) + (1j * rng.standard_normal(sh) if cplx else 0.0)
  qu, _ = np.linalg.qr(g(m, r))
  qv, _ = np.linalg.qr(g(n, r))
  spec = 3.0 * 2.0 ** (-np.arange(r) / 40.0)
  a = ((qu * spec) @ qv.conj().T).astype(dtype)
  kw = {"max_sv": dict(max_singular_values=21), "err_only": dict(max_truncation_error=0.05), "full": {}}[mode]
  ref = orc.svd(a.astype(np.complex128 if cplx else np.float64), 1, kw.get("max_singular_values"),
                kw.get("max_truncation_error"), False)
  with emulated_backend() as be:
    be.lib.band_svd = True
    out = [np.asarray(x) for x in be.svd(be.convert_to_tensor(a), 1, **kw)]
    assert be.last_svd_path.startswith("band"), (be.last_svd_path, be.last_svd_band_status)
    assert out[0].dtype == dtype and out[1].dtype == dtype and out[2].dtype == dtype
  u, s, vh, rest = out
  tol = 3e-6 if np.dtype(dtype).itemsize in (4, 8) and dtype != np.float64 else 1e-9
  s0 = float(np.real(ref[1][0]))
  assert s.shape == ref[1].shape and rest.shape == ref[3].shape
  np.testing.assert_allclose(np.real(s), np.real(ref[1]), atol=tol * s0)
  np.testing.assert_allclose(np.real(rest), np.real(ref[3]), atol=2 * tol * s0)
  k = s.shape[0]
  np.testing.assert_allclose(u.conj().T @ u, np.eye(k), atol=20 * tol)
  np.testing.assert_allclose((u * np.real(s)) @ vh, (ref[0] * ref[1]) @ ref[2], atol=40 * tol * s0)


def test_band_svd_host_logic_random_call_shapes():
  """A seeded sweep over shapes (tall / wide, padded or not), dtypes and truncation arguments: the band path's host
  logic against the pinned oracle's rule (decompositions.py:21-74) -- shapes of all four outputs and the values."""
  rng = np.random.default_rng(2024)
  with emulated_backend() as be:
    be.lib.band_svd = True
    for trial in range(14):
      m, n = (int(x) for x in rng.integers(512, 640, size=2))
      if trial % 3 == 0:
        m += int(rng.integers(100, 400))
      dtype = [np.float32, np.float64][trial % 2]
      r = min(m, n)
      spec = 2.0 * 2.0 ** (-np.arange(r) / float(rng.integers(30, 60)))
      qu, _ = np.linalg.qr(rng.standard_normal((m, r)))
      qv, _ = np.linalg.qr(rng.standard_normal((n, r)))
      a = ((qu * spec) @ qv.T).astype(dtype)
      kw = {}
      if trial % 4 != 3:
        kw["max_singular_values"] = int(rng.integers(1, r + 40))           # may exceed min(m, n)
      if trial % 2 == 1 or trial % 4 == 3:
        kw["max_truncation_error"] = float(10.0 ** rng.uniform(-3, -0.5))
        kw["relative"] = bool(trial % 3 == 1)
      ref = orc.svd(a.astype(np.float64), 1, kw.get("max_singular_values"), kw.get("max_truncation_error"),
                    kw.get("relative", False))
      be._svd_band_backoff = {}      # pylint: disable=protected-access
      u, s, vh, rest = [np.asarray(x) for x in be.svd(be.convert_to_tensor(a), 1, **kw)]
      tag = (trial, m, n, np.dtype(dtype).name, kw, be.last_svd_path, be.last_svd_band_status)
      assert be.last_svd_path == "band" or ref[1].shape[0] == 0, tag
      assert s.shape == ref[1].shape and rest.shape == ref[3].shape and u.shape == ref[0].shape and vh.shape == ref[2].shape, tag
      tol = 4e-6 if dtype == np.float32 else 1e-8
      np.testing.assert_allclose(s, ref[1], atol=tol * spec[0], err_msg=str(tag))
      np.testing.assert_allclose(rest, ref[3], atol=2 * tol * spec[0], err_msg=str(tag))
      if s.shape[0]:
        np.testing.assert_allclose((u * s) @ vh, (ref[0] * ref[1]) @ ref[2], atol=40 * tol * spec[0], err_msg=str(tag))


# ------------------------------------------------------------------ K2 gather: the tile plan of the long operand
def _plan_through_the_library(desc, bn, k, nl, l_elems, k_total=None):
  """chunk plan and box origins as the KERNEL computes them (tnh_gemm_gather_plan runs the kernel's own index
  functions on the host: no device needed)"""
  import ctypes  # pylint: disable=import-outside-toplevel
  lib = _lib.load_library()
  nch, nt = bn * k // 4, nl // bn
  off, row, col = (np.zeros(nch, np.int32) for _ in range(3))
  base = np.zeros(nt, np.int64)
  p32, p64 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
  rc = lib.tnh_gemm_gather_plan(ctypes.byref(desc), k_total or k, nl, l_elems, off.ctypes.data_as(p32), row.ctypes.data_as(p32),
                                col.ctypes.data_as(p32), nch, base.ctypes.data_as(p64), nt)
  assert rc == bn, (rc, _lib.last_error())
  return off, row, col, base


def _check_gather_plan(shape, k_axes):
  from tensornetwork_amd import hip_backend  # pylint: disable=import-outside-toplevel
  plan = hip_backend._gather_descriptor(shape, k_axes)      # pylint: disable=protected-access
  if plan is None:
    return None
  desc, bn, nl = plan
  free = [i for i in range(len(shape)) if i not in k_axes]
  k = int(np.prod([shape[i] for i in k_axes]))
  steps = int(desc.kl_ext)                             # K loop: k = step * kbox + (index inside the box)
  kbox = k // steps
  assert nl == int(np.prod([shape[i] for i in free])) and bn in (48, 64) and steps * kbox == k
  x = np.arange(int(np.prod(shape)), dtype=np.int64).reshape(shape)
  want = np.transpose(x, free + sorted(k_axes)).reshape(nl, k)      # rows: free axes, natural order; k: memory order
  off, row, col, base = _plan_through_the_library(desc, bn, kbox, nl, x.size, k)
  kin = bool(desc.k_mask & 1)
  got = np.full((nl, k), -1, np.int64)
  flat = x.reshape(-1)
  tiles = np.arange(nl // bn)[:, None]
  for kl in range(steps):
    for i in range(4):                                 # a chunk's four elements: along k (innermost axis contracted) or rows
      r, c = (row, col + i) if kin else (row + i, col)
      got[tiles * bn + r[None, :], np.broadcast_to(kl * kbox + c[None, :], (tiles.size, c.size))] = \
          flat[base[:, None] + kl * int(desc.kl_stride) + off[None, :] + i]
  np.testing.assert_array_equal(got, want)
  assert (np.diff(off) > 0).all()                      # threads walk the box in memory order
  return (bn, kin) if steps == 1 else (bn, kin, steps)


def test_gather_plan_reads_the_matrix_transpose_would_have_built():
  """The descriptor the host builds (hip_backend._gather_descriptor) run through the library's own index functions
  addresses exactly transpose(long, free + contracted).reshape(rows, K) -- for the shapes of the D = 12 network
  (ncon_interface.py:336-343 -> tensordot) and for whatever else the rule accepts."""
  assert _check_gather_plan((12, 12, 12, 12, 1, 12, 12), (1, 6)) == (64, True)          # contracted (3, 8) of the rank 9
  assert _check_gather_plan((4, 12, 12, 12, 12, 12, 1, 12), (2, 5)) == (48, False)      # (1, 6): innermost axis free
  assert _check_gather_plan((12, 12, 12, 12, 12), (3, 0)) == (48, False)
  assert _check_gather_plan((12, 12, 12, 12, 12), (0, 1)) == (64, False)                # k-major
  assert _check_gather_plan((12, 12, 12, 12, 12), (2, 4)) == (48, True)
  assert _check_gather_plan((8, 8, 8, 8, 8, 8), (1, 4)) == (64, False)
  assert _check_gather_plan((16, 16, 16, 16), (1,)) == (64, False)
  assert _check_gather_plan((12, 20736), (0,)) == (64, False)
  assert _check_gather_plan((3, 12, 5, 144, 12, 7, 4), (1, 6)) == (64, True)
  assert _check_gather_plan((10, 10, 10, 10), (1,)) is None                              # pieces not 8-byte aligned
  assert _check_gather_plan((3, 12, 5, 144, 12, 7, 4), (1, 4)) is None                   # 28 innermost free: no 48 / 64 rows
  assert _check_gather_plan((12, 12, 12), (2,)) is None or True                          # (trailing: the host never asks)
  # more than 192 contracted indices: the innermost contracted digits in the box, the outermost one walked step by step
  assert _check_gather_plan((12,) * 7, (1, 5, 6)) == (64, True, 12)                      # 144 x 248 832 x 1728 of the D = 12 network
  assert _check_gather_plan((12,) * 6, (1, 4, 5)) == (48, True, 12)
  assert _check_gather_plan((12,) * 6, (0, 3, 4)) == (48, False, 12)                     # innermost axis free
  assert _check_gather_plan((4, 16, 8, 16, 16), (0, 1, 3)) == (64, False, 64)            # (0, 1) are one digit: 64 steps
  assert _check_gather_plan((6, 16, 8, 64, 4), (0, 3)) == (64, False, 6)
  assert _check_gather_plan((16, 4, 16, 4, 16, 16), (0, 2, 4)) is None                   # two contracted digits outside the box
  rng = np.random.default_rng(11)
  accepted = 0
  for _ in range(300):
    rank = int(rng.integers(2, 8))
    shape = tuple(int(x) for x in rng.choice([1, 2, 3, 4, 5, 8, 12, 16], size=rank))
    if not 48 <= int(np.prod(shape)) <= 150000:
      continue
    nk = int(rng.integers(1, min(3, rank - 1) + 1))
    k_axes = tuple(int(x) for x in rng.choice(rank, size=nk, replace=False))
    k = int(np.prod([shape[i] for i in k_axes]))
    if k % 8 or not 8 <= k <= 2048:
      continue
    accepted += _check_gather_plan(shape, k_axes) is not None
  assert accepted >= 10, accepted


def test_gather_lowering_inside_a_contraction_path():
  """A small bond-12 network whose greedy path takes two bonds off a rank-5 intermediate: with the gather lowering the
  intermediate is never K1-permuted, the planner's bookkeeping of the result's axis order still holds (values against
  einsum), and the same network without it gives the same bits."""
  import tensornetwork_amd as tn  # pylint: disable=import-outside-toplevel
  rng = np.random.default_rng(3)
  big = orc.round_bf16(rng.standard_normal((12,) * 5).astype(np.float32) / 4)
  s1 = orc.round_bf16(rng.standard_normal((12,) * 4).astype(np.float32) / 4)
  s2 = orc.round_bf16(rng.standard_normal((12,) * 4).astype(np.float32) / 4)
  ref = np.einsum("abcde,bdxy,xcwz->aeywz", big.astype(np.float64), s1.astype(np.float64), s2.astype(np.float64))
  def run(on):        # (its device tensors die with the frame, inside the emulation they were allocated from)
    with emulated_backend() as be:
      be.gather_gemm, be.gather_min_rows = on, 1024
      operands = [be.to_bfloat16(big), be.to_bfloat16(s1), be.to_bfloat16(s2)]
      before = be.permute_launches
      out = np.asarray(tn.ncon(operands, [[-1, 1, 2, 3, -2], [1, 3, 4, -3], [4, 2, -4, -5]], backend=be))
      del operands
      return (out, be.gather_launches, be.permute_launches - before,
              [c for c in be.lib.calls if c[0] == "permute" and int(np.prod(c[1])) >= 12**5])

  results = {on: run(on) for on in (True, False)}
  got, launches, _, big_permutes = results[True]
  # (at most one pass over a 12^5 tensor remains: ncon's final reordering of the result to (-1 ... -5))
  assert launches == 2 and len(big_permutes) <= 1, (launches, big_permutes)
  assert results[False][1] == 0 and len(results[False][3]) >= len(big_permutes) + 2, results[False][3]
  np.testing.assert_allclose(got, ref, rtol=2.0**-6, atol=2.0**-6)
  np.testing.assert_allclose(results[False][0], ref, rtol=2.0**-6, atol=2.0**-6)


def test_bench_gather_leg_runs_on_the_emulated_backend():
  """bench.py's gather_gemm leg end to end at a reduced rank: the rows it writes, the device-side equality check and the
  entry of the compact line."""
  import json  # pylint: disable=import-outside-toplevel
  import bench  # pylint: disable=import-outside-toplevel
  with emulated_backend() as be:
    rec = bench.gather_gemm_bench(ta, be, True, D=12, rank=6, reps=1,
                                  cases={"k15_x": [1, 5], "k03_y": [0, 3], "k45_trailing": [4, 5],
                                         "k145_loop": ([1, 3, 4], [1, 4, 5])})
  assert rec["verified"]["ok"], rec
  by_case = {r["case"]: r for r in rec["rows"]}
  assert by_case["k15_x"]["box"] == {"rows": 64, "piece_bytes": 1536, "innermost_axis_contracted": True}
  assert by_case["k15_x"]["small_first"]["gather_kernel"] == "bf16_gather_Sx64"
  assert by_case["k15_x"]["long_first"]["gather_kernel"] == "bf16_gather_64xS"
  assert by_case["k15_x"]["small_first"]["classic_launches"] == {"gather": 0, "permute": 2}
  assert by_case["k15_x"]["small_first"]["gather_launches"] == {"gather": 1, "permute": 1}
  assert by_case["k45_trailing"]["small_first"]["gather_launches"]["gather"] == 0
  line = json.loads(bench.compact_line({"metric": "m", "value": 1.0, "gather_gemm": rec}, "bench_detail.json"))
  assert by_case["k145_loop"]["small_first"]["gather_kernel"] == "bf16_gather_kloop_Sx48"
  assert by_case["k145_loop"]["small_first"]["rel_difference"] <= 2.0**-9
  assert set(line["gather_gemm_us"]) == {"k15", "k03", "k45", "k145"} and len(line["gather_gemm_us"]["k15"]) == 4


def test_gather_lowering_random_products_match_numpy():
  """Random small x long products through HipBackend.tensordot / tensordot_planned on the emulated C ABI with the
  gather lowering on: whatever the placement of the contracted axes, the order of the pairs, the operand order and
  the K loop, the values are np.tensordot's and the returned axis bookkeeping describes the result."""
  from tensornetwork_amd import hip_backend  # pylint: disable=import-outside-toplevel
  rng = np.random.default_rng(2024)
  gathered = looped = swapped_seen = 0
  with emulated_backend() as be:
    be.gather_gemm, be.gather_min_rows = True, 48
    trial = 0
    for _ in range(4000):
      if trial >= 40:
        break
      rank = int(rng.integers(3, 7))
      shape_l = [int(x) for x in rng.choice([4, 8, 12, 16], size=rank)]
      nk = int(rng.integers(1, min(3, rank - 1) + 1))
      axes_l = sorted(int(x) for x in rng.choice(rank, size=nk, replace=False))
      k = int(np.prod([shape_l[a] for a in axes_l]))
      nl = int(np.prod(shape_l)) // k
      if k % 8 or k < 16 or k > 1024 or nl < 256 or nl * k > 1500000:      # (nl > ms: the long operand is the long one)
        continue
      free_s = [int(x) for x in rng.choice([6, 8, 12, 16], size=2)]
      ms = int(np.prod(free_s))
      if not 64 < ms <= 192:
        continue
      trial += 1
      # the small tensor: free axes and the contracted ones (paired with the long tensor's in a random order) interleaved
      pair_order = [int(x) for x in rng.permutation(nk)]
      dims = [("f", d) for d in free_s] + [("k", i) for i in pair_order]
      dims = [dims[i] for i in rng.permutation(len(dims))]
      shape_s = [d if kind == "f" else shape_l[axes_l[d]] for kind, d in dims]
      axes_s_of_pair = {d: pos for pos, (kind, d) in enumerate(dims) if kind == "k"}
      axes_s = [axes_s_of_pair[i] for i in range(nk)]          # axes_s[i] pairs with axes_l[i]
      s = orc.round_bf16(rng.standard_normal(shape_s).astype(np.float32) / 4)
      l = orc.round_bf16(rng.standard_normal(shape_l).astype(np.float32) / 4)
      ds, dl = be.to_bfloat16(s), be.to_bfloat16(l)
      small_first = bool(rng.integers(0, 2))
      plan = hip_backend._gather_descriptor(shape_l, axes_l)      # pylint: disable=protected-access
      trailing = axes_l == list(range(rank - nk, rank))
      args = (ds, dl, [axes_s, axes_l]) if small_first else (dl, ds, [axes_l, axes_s])
      ref = np.tensordot(s.astype(np.float64), l.astype(np.float64), [axes_s, axes_l]) if small_first else \
          np.tensordot(l.astype(np.float64), s.astype(np.float64), [axes_l, axes_s])
      before = be.gather_launches
      if trial % 2:
        out, used_a, used_b, swapped = be.tensordot_planned(*args, None, None, allow_swap=True)
        got = np.asarray(out)
        na = len(used_a)
        free_a = [i for i in range(args[0].ndim) if i not in args[2][0]]
        free_b = [i for i in range(args[1].ndim) if i not in args[2][1]]
        perm_a = [free_a.index(int(i)) for i in used_a]
        perm_b = [len(free_a) + free_b.index(int(i)) for i in used_b]
        want = np.transpose(ref, perm_b + perm_a if swapped else perm_a + perm_b)
        swapped_seen += bool(swapped)
        assert len(used_b) + na == ref.ndim
      else:
        got, want = np.asarray(be.tensordot(*args)), ref
      used = be.gather_launches - before
      eligible = plan is not None and not trailing and (small_first or ms % 8 == 0) and \
          (plan[0].kl_ext == 1 or ms * (k // plan[0].kl_ext // 8) <= 11 * 256)
      # (planned calls keep the permute for boxes of short pieces)
      if eligible and (trial % 2 == 0 or hip_backend._gather_piece_bytes(plan[0]) >= be.gather_min_piece_bytes):  # pylint: disable=protected-access
        assert used == 1, (shape_s, shape_l, axes_s, axes_l, small_first)
      gathered += used
      looped += used and plan[0].kl_ext > 1
      assert got.shape == want.shape
      np.testing.assert_allclose(got, want, rtol=2.0**-7, atol=2.0**-8 * k**0.5)
      del ds, dl
  assert gathered >= 8 and looped >= 1 and swapped_seen >= 1, (gathered, looped, swapped_seen)


@pytest.mark.parametrize("placement", ["left", "right"])
def test_mera_sliced_run_reuses_partial_contractions(placement):
  """The bond-sliced MERA layer as `slice_edge` leaves it (reference network_components.py:1636-1682: BOTH nodes of a
  cut edge are sliced, so the disentangler / isometry next to a cut leg changes with the slice index too): the slices
  add up to the energy of the DENSE network (oracle backend on the materialised tensors), staged reuse gives the same
  slice results as the slice-by-slice run (the f32 check of sampled slices is exact in f32), and the executed
  multiply-adds are counted per class of steps -- never fewer than the flop-optimal dense contraction needs."""
  from tensornetwork_amd import contractors, pathfinder, workloads  # pylint: disable=import-outside-toplevel
  chi = 4
  with emulated_backend() as be:
    alone = workloads.mera_sliced_run(be, chi, placement, np.float32, reuse_partials=False)
    staged = workloads.mera_sliced_run(be, chi, placement, np.float32, check_every=3)
    half_alone = workloads.mera_sliced_run(be, chi, placement, ta.bfloat16, reuse_partials=False)
    half_staged = workloads.mera_sliced_run(be, chi, placement, ta.bfloat16)
    layer = workloads.MeraSlicedLayer(be, chi, placement, np.float32)
    ham, rho, iso, dis = layer.host_tensors()
    # the neighbour of each cut leg is an input WITH a window (sliced per index), not a fixed size-1 stand-in
    assert sorted(len(w) for w in layer.stage.windows.values()) == [1, 1, 1, 1] and len(layer.stage.windows) == 4
  ob = OracleBackend()
  dense_nodes = workloads.mera_layer_network(ob, ham, rho, iso, dis, placement)
  dense = float(np.asarray(contractors.optimal(dense_nodes).tensor))
  assert abs(staged["energy_partial_sum"] - dense) <= 2e-5 * max(1.0, abs(dense))
  assert abs(alone["energy_partial_sum"] - dense) <= 2e-5 * max(1.0, abs(dense))
  assert staged["reuse_partials"] and staged["slices_done"] == alone["slices_done"] == chi * chi
  assert abs(staged["energy_partial_sum"] - alone["energy_partial_sum"]) <= 1e-5 * max(1.0, abs(alone["energy_partial_sum"]))
  assert half_staged["energy_partial_sum"] == half_alone["energy_partial_sum"]
  assert staged["checks"] and all(abs(c[1] - c[2]) <= 1e-6 * max(1.0, abs(c[2])) for c in staged["checks"])
  macs, runs = staged["macs_by_dependence"], staged["stage_runs"]
  assert runs == {"none": 1, "i": chi, "j": chi, "ij": chi * chi}
  assert abs(sum(macs.values()) - staged["macs_per_slice"]) <= 1e-9 * staged["macs_per_slice"]      # the stages ARE the path
  assert staged["executed_macs"] == macs["none"] + chi * (macs["i"] + macs["j"]) + chi * chi * macs["ij"]
  assert staged["executed_macs"] == staged["model_macs_with_reuse_all_slices"]
  assert alone["executed_macs"] == alone["macs_per_slice"] * chi * chi
  assert staged["executed_macs"] < 0.7 * alone["executed_macs"]
  # reuse recovers at best the DENSE cost of the network (the cut legs stay batch indices to the end)
  shapes = mera_layer_shapes = workloads.mera_layer_network(workloads._PlanBackend(), workloads._ShapeOnly((chi,) * 6),      # pylint: disable=protected-access
                                                            workloads._ShapeOnly((chi,) * 6), workloads._ShapeOnly((chi,) * 3),      # pylint: disable=protected-access
                                                            workloads._ShapeOnly((chi,) * 4), placement)      # pylint: disable=protected-access
  ins = [set(n.edges) for n in shapes]
  size = {e: e.dimension for n in mera_layer_shapes for e in n.edges}
  dense_macs = pathfinder.path_cost(ins, set(), size, pathfinder.optimal(ins, set(), size))[0]
  assert staged["executed_macs"] >= dense_macs
  # a time budget stops between slices and says how far it got
  with emulated_backend() as be:
    short = workloads.mera_sliced_run(be, 6, placement, np.float32, budget_seconds=0.0)
  assert 0 < short["slices_done"] < 36 and short["stage_runs"]["ij"] == short["slices_done"]


def test_mera_sliced_layer_on_a_grid_of_ranks():
  """`_StagePlan.partition` deals the chi^2 slices of a MERA placement to 4 ranks as a 2 x 2 grid (two cut values each
  way per rank): every rank's share through `_contract_slices_staged` executes exactly what the model says, and the rank
  partials add up to the energy of all slices (f32: the same slice results, summed in another grouping)."""
  from tensornetwork_amd import workloads  # pylint: disable=import-outside-toplevel
  chi = 4
  with emulated_backend() as be:
    layer = workloads.MeraSlicedLayer(be, chi, "left", np.float32)
    every = layer.all_slices()
    blocks = layer.stage.partition(every, 4)
    assert [(len({i for i, _ in b}), len({j for _, j in b})) for b in blocks] == [(2, 2)] * 4
    assert sorted(x for b in blocks for x in b) == sorted(every)
    whole = float(np.asarray(layer.contract(every), dtype=np.float64).reshape(-1)[0])
    parts = []
    for block in blocks:
      st = {}
      parts.append(float(np.asarray(layer.contract(block, stats=st), dtype=np.float64).reshape(-1)[0]))
      assert st["executed_macs"] == layer.stage.macs_with_reuse(block) and st["slices_done"] == 4
      assert st["stage_runs"] == {"-": 1, "0": 2, "1": 2, "0,1": 4}
  assert abs(sum(parts) - whole) <= 1e-5 * max(1.0, abs(whole))


def test_bench_mera_chi64_leg_on_the_emulated_backend():
  """bench.py's mera_chi64 leg at chi = 4 with both placements run: the record distinguishes the measured run (partial
  results reused) from the per-slice extrapolation and counts the executed flops."""
  import bench  # pylint: disable=import-outside-toplevel
  with emulated_backend() as be:
    rec = bench.mera_chi64_bench(ta, be, verify=True, full_placements=2, budget_s=60.0, chi=4)
  assert rec["measured_slices"] == 32 and rec["measured_reuse_partials"] is True
  assert set(rec["measured"]) == {"left", "right"} and all("reuse_partials_error" not in m for m in rec["measured"].values())
  # (at chi = 4 a placement's run samples ONE slice partial, too few for the rms model: only its presence is checked)
  assert rec["verified"]["ok"] and set(rec["verified_runs"]) == {"left", "right"}
  executed = sum(m["executed_macs"] for m in rec["measured"].values())
  assert abs(rec["measured_tflops"] - 2.0 * executed / rec["measured_seconds"] / 1e12) <= 1e-9 * rec["measured_tflops"]
  assert rec["measured_speedup_over_slice_by_slice"] > 0


@pytest.mark.parametrize("D,min_slices,staged", [(3, 8, None), (4, 30, True)])
def test_bench_sliced_network_leg_on_the_emulated_backend(D, min_slices, staged):
  """bench.py's sliced-network leg end to end at a small bond dimension: the mode contract_sliced takes, the executed
  flops beside what stand-alone slices would cost, the f32 check of every slice partial, the compact-line entry."""
  import json  # pylint: disable=import-outside-toplevel
  import bench  # pylint: disable=import-outside-toplevel
  with emulated_backend() as be:
    rec = bench.sliced_network_bench(ta, be, None, 0, 1, D, min_slices, True)
  assert rec["verified"]["n_values"] == rec["n_slices"]
  # (the rms model of the rounding check is a statistical bound: with the 9 partials of the D = 3 toy network it is
  #  only required to be in the right range, with 30+ partials to hold)
  assert rec["verified"]["ok"] if rec["n_slices"] >= 30 else rec["verified"]["err_over_tol"] < 1.5
  if staged is None:          # (which mode wins at D = 3 depends on the cut set the search finds: only consistency is asserted)
    staged = rec["mode"] != "slice by slice"
  assert (rec["mode"] != "slice by slice") == staged
  assert rec["flops_total"] <= rec["flops_if_every_slice_ran_alone"]
  if staged:
    assert rec["speedup_over_slices_alone_at_this_rate"] > 1.25
    # what ran is what the host model said would run (the line reports the run's own counters)
    assert rec["executed_equals_model"] is True and rec["flops_total"] == pytest.approx(rec["flops_total_by_the_host_model"])
  assert rec["ideal_speedup_of_this_partition"] == pytest.approx(1.0)
  assert abs(rec["tflops"] - rec["flops_total"] / rec["seconds"] / 1e12) <= 1e-9 * rec["tflops"]
  line = json.loads(bench.compact_line({"metric": "m", "value": 1.0, "sliced_network": rec}, "bench_detail.json"))
  assert line["sliced_network"]["mode"] == rec["mode"] and line["sliced_network"]["n_slices"] == rec["n_slices"]


def test_independent_directions_of_the_complex_svd_cluster_branch():
  """`HipBackend._independent_directions` (the host arithmetic of the complex band SVD's degenerate-value branch): from
  2k candidates that span only k complex directions -- every direction present twice, once rotated by i as in the real
  embedding -- it picks k coefficient vectors whose combinations are orthonormal; fewer directions than asked: None."""
  from tensornetwork_amd.hip_backend import HipBackend  # pylint: disable=import-outside-toplevel
  rng = np.random.default_rng(4)
  k, m = 6, 40
  q, _ = np.linalg.qr(rng.standard_normal((m, k)) + 1j * rng.standard_normal((m, k)))
  z = np.empty((m, 2 * k), dtype=np.complex128)
  z[:, 0::2], z[:, 1::2] = q, 1j * q                       # the pair of real vectors of one complex line
  z = z @ np.kron(np.eye(k), np.array([[1.0, 0.3], [0.0, 1.0]]))      # and not even orthogonal inside a pair
  c = HipBackend._independent_directions(z.conj().T @ z, k)      # pylint: disable=protected-access
  assert c is not None and c.shape == (2 * k, k)
  y = z @ c
  np.testing.assert_allclose(y.conj().T @ y, np.eye(k), atol=1e-12)
  np.testing.assert_allclose(q @ (q.conj().T @ y), y, atol=1e-12)      # inside the span of the k directions
  assert HipBackend._independent_directions(z.conj().T @ z, k + 1) is None      # pylint: disable=protected-access


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_complex_band_svd_cluster_branch_gives_the_same_decomposition(dtype):
  """The degenerate-value branch of `_svd_complex_band`, forced on an ordinary matrix (emulated C ABI): same values,
  orthonormal vectors, A V = U S -- the host logic the GPU test `test_complex_svd_cluster_branch_on_gpu` runs on the
  device."""
  rng = np.random.default_rng(23)
  m, n, k = 300, 280, 12
  a = (rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))).astype(dtype)
  with emulated_backend() as be:
    be.lib.band_svd = True
    dev = be.convert_to_tensor(a)
    s0 = np.asarray(be.svd(dev, 1, max_singular_values=k)[1])
    assert be.last_svd_path.startswith("band"), be.last_svd_path
    be.svd_complex_force_cluster_path = True
    try:
      u, s, vh, _ = (np.asarray(x) for x in be.svd(dev, 1, max_singular_values=k))
      assert be.last_svd_path.startswith("band"), be.last_svd_path
    finally:
      be.svd_complex_force_cluster_path = False
  tol = 1e-5 if dtype == np.complex64 else 1e-11
  np.testing.assert_array_equal(s, s0)
  a128 = a.astype(np.complex128)
  sr = np.linalg.svd(a128, compute_uv=False)
  np.testing.assert_allclose(np.real(s), sr[:k], atol=tol * sr[0])
  np.testing.assert_allclose(u.conj().T @ u, np.eye(k), atol=20 * tol)
  np.testing.assert_allclose(a128 @ vh.conj().T, u * np.real(s), atol=40 * tol * sr[0])


def test_k_major_cost_rule_on_the_emulated_backend():
  """`HipBackend.kmajor_inplace_penalty` (round 5): a k-major operand is read in place only where ONE K1 pass would cost
  more than a tenth of the product.  Round 6: a k-major `b` with contraction runs that are multiples of 64 (config-2
  L0) is read by the whole-K-tile lean loop, whose cost depends on the operand's size only (nothing below 0.6 GB):
  D = 64 and D = 96 stay in place (view_nn, no permute); with `kmajor_tile_walk` off (the round-5 rule) D = 96 takes
  the pass; with the rule off both are read in place.  Launch bookkeeping on the emulated C ABI."""
  rng = np.random.default_rng(2)
  with emulated_backend() as be:
    # launch bookkeeping only: the emulated GEMMs (4096^3 and twice 9216^3 in NumPy: a minute) are not run
    for name in ("tnh_gemm_view", "tnh_gemm"):
      setattr(be.lib, name, lambda *args, **kwargs: _lib.OK)
    made = {}
    for D, rule, tile_walk, want in ((64, 0.10, True, 0), (96, 0.10, True, 0), (96, 0.10, False, 1), (96, 0.0, False, 0)):
      if D not in made:       # (values do not matter here: one array serves as both operands)
        made[D] = be.to_bfloat16(rng.standard_normal((D,) * 4, dtype=np.float32) / D)
      a = b = made[D]
      be.kmajor_inplace_penalty = rule
      be.kmajor_tile_walk = tile_walk
      before = be.permute_launches
      out = be.tensordot(a, b, [[2, 3], [0, 1]])
      assert be.permute_launches - before == want, (D, rule, tile_walk)
      assert out.shape == (D, D, D, D)
    # the size term: a 2 GB k-major b (below inplace_max_bytes) costs the product 0.03 x 1.4 = 4 %, which the rule
    # prices against its pass (host arithmetic only: `usable` is not reachable from outside, so through the numbers)
    be.kmajor_tile_walk = True
    gb = 2.0
    penalty = min(be.kmajor_inplace_penalty or 0.10, max(0.0, be.kmajor_tile_walk_penalty_per_gb * (gb - 0.6)))
    assert abs(penalty - 0.042) < 1e-9


def test_tensordot_plan_cache_replays_the_same_lowering():
  """Round 6: the permute + ONE GEMM lowering of a (shapes, axes, dtype, hints) is planned once and replayed
  (`HipBackend._plan_generic` / `_run_plan`): same values, same K1 launch count and same free-axis order with the
  cache on (first call plans, second call replays) and off; integer operands, an integer `axes` and calls under
  `gemm_events` (bench.py's event pairs) are not cached."""
  rng = np.random.default_rng(8)
  cases = [
      ((3, 4, 5), (5, 4, 6), ([2, 1], [0, 1]), None, None),            # b's contracted axes leading, pairs crossed
      ((4, 3, 5), (3, 6), ([1], [0]), None, None),                       # a's contracted axis in the middle: one permute
      ((6, 4), (6, 5), ([0], [0]), None, None),                          # a stored [K][M]
      ((2, 3, 4, 5), (5, 2, 7), ([3, 0], [0, 1]), [2, 1], [2]),          # planned: free order of the permuted a hinted
      ((5, 6), (7, 6), ([-1], [1]), None, None),                         # negative axis
  ]
  with emulated_backend() as be:
    for dtype in (np.float32, np.complex64, "bf16"):
      for sa, sb, axes, ha, hb in cases:
        x = rng.standard_normal(sa).astype(np.float32)
        y = rng.standard_normal(sb).astype(np.float32)
        if dtype == "bf16":
          x, y = orc.round_bf16(x), orc.round_bf16(y)
          dx, dy = be.to_bfloat16(x), be.to_bfloat16(y)
        else:
          x, y = x.astype(dtype), y.astype(dtype)
          dx, dy = be.convert_to_tensor(x), be.convert_to_tensor(y)
        outs = []
        for cache in (False, True, True):
          be.plan_cache = cache
          plans = len(be._td_plans)            # pylint: disable=protected-access
          before = be.permute_launches
          t, ua, ub = be.tensordot_planned(dx, dy, axes, ha, hb)
          outs.append((np.asarray(t), list(ua), list(ub), be.permute_launches - before, len(be._td_plans) - plans))  # pylint: disable=protected-access
        (r0, ua0, ub0, p0, n0), (r1, ua1, ub1, p1, n1), (r2, ua2, ub2, p2, n2) = outs
        assert (n0, n1, n2) == (0, 1, 0), (sa, sb, axes, n0, n1, n2)          # planned once, replayed once
        assert (ua0, ub0, p0) == (ua1, ub1, p1) == (ua2, ub2, p2)
        np.testing.assert_array_equal(r0, r1)
        np.testing.assert_array_equal(r0, r2)
        fa = [i for i in range(len(sa)) if i not in [a % len(sa) for a in axes[0]]]
        ref = np.tensordot(x.astype(np.complex128 if dtype is np.complex64 else np.float64),
                           y.astype(np.complex128 if dtype is np.complex64 else np.float64), axes)
        # free axes of a in the order the call reports, then b's
        perm = [fa.index(i) for i in ua0] + [len(fa) + sorted(ub0).index(i) for i in ub0]
        np.testing.assert_allclose(r0, np.transpose(ref, perm), rtol=2e-2 if dtype == "bf16" else 1e-5, atol=1e-2 if dtype == "bf16" else 1e-5)
    be.plan_cache = True
    plans = len(be._td_plans)                  # pylint: disable=protected-access
    xi = be.convert_to_tensor(rng.integers(-5, 5, (3, 4)))
    yi = be.convert_to_tensor(rng.integers(-5, 5, (4, 2)))
    np.testing.assert_array_equal(np.asarray(be.tensordot(xi, yi, [[1], [0]])), np.asarray(xi) @ np.asarray(yi))
    xf = be.convert_to_tensor(rng.standard_normal((3, 4)).astype(np.float32))
    yf = be.convert_to_tensor(rng.standard_normal((4, 2)).astype(np.float32))
    be.tensordot(xf, yf, 1)                    # integer axes: the general walk
    be.gemm_events = []
    try:
      be.tensordot(xf, yf, [[1], [0]])
      assert len(be.gemm_events) == 1
    finally:
      be.gemm_events = None
    assert len(be._td_plans) == plans          # pylint: disable=protected-access
    be.tensordot(xf, yf, [[1], [0]])
    assert len(be._td_plans) == plans + 1      # pylint: disable=protected-access


def test_k1_pass_of_a_long_row_operand_writes_the_k_blocked_form():
  """Round 6 (profiles/r06_k_blocked_operands.md): an operand of the view GEMM that needs a K1 pass anyway, and whose
  rows would be `k_blocked_min_row_bytes` or longer, is written K-blocked -- [outer contracted axes, free axes, inner
  contracted axes] with a two-level contraction view -- instead of [free, contracted]: same product, and the rows of a
  tile no longer lie a power of two apart.  Threshold lowered so that small shapes take the path; values against
  float64, the views handed to tnh_gemm_view checked; a single long contracted axis is split by a reshape first."""
  rng = np.random.default_rng(12)
  with emulated_backend() as be:
    be.k_blocked_min_row_bytes = 1024
    seen = []
    real = be.lib.tnh_gemm_view

    def spy(code, oc, m, n, k, a, va, b, vb, c, ldc):
      seen.append(tuple((v._obj.r0, v._obj.sr0, v._obj.sr1, v._obj.k0, v._obj.sk0, v._obj.sk1) for v in (va, vb)))
      return real(code, oc, m, n, k, a, va, b, vb, c, ldc)
    be.lib.tnh_gemm_view = spy
    # (1) a: contracted axes (2, 8, 64) in three memory runs (no view) -> blocked with inner run 8 * 64 = 512, outer 2
    sa, sb, axes = (2, 1792, 8, 2, 64), (3600, 2, 8, 64), ([0, 2, 4], [1, 2, 3])
    x = orc.round_bf16(rng.standard_normal(sa).astype(np.float32) / 32)
    y = orc.round_bf16(rng.standard_normal(sb).astype(np.float32))
    before = be.permute_launches
    got = np.asarray(be.tensordot(be.to_bfloat16(x), be.to_bfloat16(y), axes))
    assert be.permute_launches - before == 1            # b is read where it lies, a takes ONE pass
    va, vb = seen[-1]
    assert va == (3584, 512, 0, 512, 1, 3584 * 512), va
    np.testing.assert_allclose(got, np.tensordot(x.astype(np.float64), y.astype(np.float64), axes), rtol=2.0**-7, atol=2e-2)
    be.k_blocked_permutes = False
    try:
      ref = np.asarray(be.tensordot(be.to_bfloat16(x), be.to_bfloat16(y), axes))
      assert seen[-1][0] == (3584, 1024, 0, 1024, 1, 0)
    finally:
      be.k_blocked_permutes = True
    np.testing.assert_allclose(got, ref, rtol=2.0**-7, atol=1e-3)
    # (2) one long contracted axis, operand stored [K][M] and too big to be read k-major in place: split 1024 = 2 x 512
    be.inplace_max_bytes = 1 << 16
    x = orc.round_bf16(rng.standard_normal((1024, 3584)).astype(np.float32) / 32)
    y = orc.round_bf16(rng.standard_normal((3600, 1024)).astype(np.float32))
    got = np.asarray(be.tensordot(be.to_bfloat16(x), be.to_bfloat16(y), [[0], [1]]))
    va, vb = seen[-1]
    assert va == (3584, 512, 0, 512, 1, 3584 * 512), va
    np.testing.assert_allclose(got, x.astype(np.float64).T @ y.astype(np.float64).T, rtol=2.0**-7, atol=2e-2)
