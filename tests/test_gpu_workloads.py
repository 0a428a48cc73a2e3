"""GPU parity tests, workload level: ncon / contract_between / split_node /
contractors through HipBackend against (a) the golden outputs of the reference,
(b) the CPU oracle on seeded inputs, (c) size-independent properties at the
BASELINE.json sizes where a full CPU check would take too long."""
import numpy as np
import pytest

import tensornetwork_amd as ta
from tensornetwork_amd import _lib, contractors, network
from oracle import numpy_oracle as orc
import cases as C

pytestmark = pytest.mark.gpu


def test_ncon_golden(hip, golden):
  for case in golden.cases["ncon"]:
    C.assert_close(C.run_ncon(hip, golden, case), golden[case["out"]])


def test_contract_between_golden(hip, golden):
  for case in golden.cases["contract_between"]:
    C.assert_close(C.run_contract_between(hip, golden, case), golden[case["out"]])


def test_split_node_golden(hip, golden):
  for case in golden.cases["split_node"]:
    left, right, trun, recon = C.run_split(hip, golden, case)
    assert list(left.shape) == case["left_shape"], case["x"]
    assert list(right.shape) == case["right_shape"], case["x"]
    x = golden[case["x"]]
    scale = float(np.max(np.abs(x))) * np.sqrt(x.size) + 1e-30
    C.assert_close(trun, golden[case["trun"]], scale=scale)
    C.assert_close(recon, golden[case["recon"]], scale=scale)


def test_contractors_golden(hip, golden):
  for case in golden.cases["contractors"]:
    C.assert_close(C.run_contractor(hip, golden, case), golden[case["out"]])


def test_config1_readme_ncon(hip):
  # BASELINE config 1 on the new backend: exact
  a = np.ones((10, 10))
  out = ta.ncon([a, a], [(-1, 1), (1, -2)], backend=hip)
  np.testing.assert_array_equal(np.asarray(out), 10 * np.ones((10, 10)))


def _config2(hip, D, layout, seed=2):
  """two rank-4 bf16 nodes, two shared bonds of dimension D."""
  rng = np.random.default_rng(seed)
  A = orc.round_bf16(rng.standard_normal((D,) * 4) / D)
  B = orc.round_bf16(rng.standard_normal((D,) * 4) / D)
  a = network.Node(hip.to_bfloat16(A), backend=hip)
  b = network.Node(hip.to_bfloat16(B), backend=hip)
  conn = [(2, 0), (3, 1)] if layout == "L0" else [(1, 2), (3, 0)]
  for x, y in conn:
    network.connect(a[x], b[y])
  out = network.contract_between(a, b)
  return A, B, conn, out


@pytest.mark.parametrize("layout", ["L0", "L1"])
@pytest.mark.parametrize("D", [16, 32])
def test_config2_bf16_contract_between_full_check(hip, D, layout):
  A, B, conn, out = _config2(hip, D, layout)
  ref = np.tensordot(A.astype(np.float64), B.astype(np.float64), [[c[0] for c in conn], [c[1] for c in conn]])
  got = np.asarray(out.tensor)
  assert got.shape == ref.shape
  # bf16 output: half an ulp (2^-9) of the value + fp32 accumulation noise
  np.testing.assert_allclose(got, ref, rtol=2.0**-8, atol=2.0**-8 * np.abs(ref).max() * 0.05 + 1e-6)
  if D >= 32:
    assert hip.lib.tnh_gemm_last_kernel().decode().startswith("bf16_nt"), "MFMA speed path not taken"


@pytest.mark.parametrize("layout", ["L0", "L1"])
def test_config2_bf16_D64_sampled_entries(hip, layout):
  D = 64
  A, B, conn, out = _config2(hip, D, layout)
  got = np.asarray(out.tensor)
  rng = np.random.default_rng(0)
  A64, B64 = A.astype(np.float64), B.astype(np.float64)
  free_a = [i for i in range(4) if i not in [c[0] for c in conn]]
  free_b = [i for i in range(4) if i not in [c[1] for c in conn]]
  for _ in range(1024):
    ia, ib = rng.integers(0, D, 2), rng.integers(0, D, 2)
    sa = [slice(None)] * 4
    sb = [slice(None)] * 4
    for ax, v in zip(free_a, ia):
      sa[ax] = int(v)
    for ax, v in zip(free_b, ib):
      sb[ax] = int(v)
    va, vb = A64[tuple(sa)], B64[tuple(sb)]     # remaining axes: the contracted ones, in axis order
    ids_a = [int(i) for i in np.argsort([c[0] for c in conn])]   # pair id carried by each axis of va
    ids_b = [int(i) for i in np.argsort([c[1] for c in conn])]
    ref = float(np.einsum(va, ids_a, vb, ids_b))
    val = got[tuple(ia) + tuple(ib)]
    assert abs(val - ref) <= 2.0**-8 * abs(ref) + 2e-4, (ia, ib, val, ref)
  assert hip.lib.tnh_gemm_last_kernel().decode().startswith(("bf16_nt", "bf16_view"))


def test_tensordot_linearity_and_identity_large(hip):
  """Size-independent properties at a GEMM size the CPU oracle would need minutes for."""
  rng = np.random.default_rng(11)
  M = N = K = 4096
  a = hip.to_bfloat16(rng.standard_normal((M, K)).astype(np.float32))
  eye = hip.cast(hip.eye(K, dtype=np.float32), ta.bfloat16)
  out = hip.tensordot(a, eye, [[1], [1]])          # A . I^T == A exactly (one nonzero product per output)
  assert hip.lib.tnh_gemm_last_kernel().decode().startswith(("bf16_nt", "bf16_view"))
  np.testing.assert_array_equal(np.asarray(out), np.asarray(a))
  # linearity in fp32: (2A).B == 2(A.B) exactly (power-of-two scaling commutes with rounding)
  b = hip.to_bfloat16(rng.standard_normal((N, K)).astype(np.float32))
  c1 = np.asarray(hip.tensordot(a * 2.0, b, [[1], [1]]))
  c2 = np.asarray(hip.tensordot(a, b, [[1], [1]]))
  np.testing.assert_array_equal(c1, 2.0 * c2)
  # a few sampled entries against fp64 dot products
  ah, bh = np.asarray(a).astype(np.float64), np.asarray(b).astype(np.float64)
  for i, j in rng.integers(0, 4096, (64, 2)):
    ref = ah[i] @ bh[j]
    assert abs(c2[i, j] - ref) <= 2.0**-8 * abs(ref) + 0.05


def test_split_node_config3_small(hip):
  """Config 3 at n = 256 (rank-6 node (4,)*6 ... scaled): mixed edge order + truncation."""
  rng = np.random.default_rng(3)
  x = rng.standard_normal((4,) * 8).astype(np.float32)  # 256 x 256 after the split
  node = network.Node(x, backend=hip)
  left_axes, right_axes = [0, 2, 4, 6], [1, 3, 5, 7]
  k = 16
  left, right, trun = network.split_node(node, [node[i] for i in left_axes], [node[i] for i in right_axes],
                                         max_singular_values=k)
  mat = np.transpose(x, left_axes + right_axes).reshape(256, 256).astype(np.float64)
  u, s, vh = np.linalg.svd(mat)
  np.testing.assert_allclose(np.asarray(trun), s[k:], atol=1e-5 * s[0])
  assert left.shape == (4, 4, 4, 4, k) and right.shape == (k, 4, 4, 4, 4)
  recon = np.asarray(network.contract_between(left, right).tensor).reshape(256, 256)
  best = (u[:, :k] * s[:k]) @ vh[:k]
  assert np.linalg.norm(recon - best) <= 1e-4 * np.linalg.norm(mat)


def test_greedy_mps_chain_and_regular_graph(hip):
  """Config-4 topology (MPS overlap) and the north-star random regular network, small bond."""
  rng = np.random.default_rng(5)
  n_sites, d, D = 10, 2, 16
  dims = [1] + [D] * (n_sites - 1) + [1]
  kets = [(rng.standard_normal((dims[i], d, dims[i + 1])) / np.sqrt(d * D)).astype(np.float32) for i in range(n_sites)]

  def build(be):
    nk = [network.Node(k, backend=be) for k in kets]
    nb = [network.Node(np.conj(k), backend=be) for k in kets]
    for i in range(n_sites):
      network.connect(nk[i][1], nb[i][1])
      if i + 1 < n_sites:
        network.connect(nk[i][2], nk[i + 1][0])
        network.connect(nb[i][2], nb[i + 1][0])
    network.connect(nk[0][0], nb[0][0])
    network.connect(nk[-1][2], nb[-1][2])
    return nk + nb

  got = np.asarray(contractors.greedy(build(hip)).tensor)
  ref = contractors.greedy(build(orc.OracleBackend())).tensor
  np.testing.assert_allclose(got, ref, rtol=1e-4)


def test_reference_library_dropin_if_available(hip):
  """When google/TensorNetwork itself is importable, drive IT with backend='hip'."""
  tn = pytest.importorskip("tensornetwork")
  assert "hip" in tn.backends.backend_factory._BACKENDS  # pylint: disable=protected-access
  rng = np.random.default_rng(1)
  a_val, b_val = rng.standard_normal((4, 5, 6)), rng.standard_normal((6, 5, 7))
  a, b = tn.Node(a_val, backend="hip"), tn.Node(b_val, backend="hip")
  a[2] ^ b[0]  # pylint: disable=pointless-statement
  a[1] ^ b[1]  # pylint: disable=pointless-statement
  c = a @ b
  np.testing.assert_allclose(np.asarray(c.tensor), np.tensordot(a_val, b_val, [[2, 1], [0, 1]]), rtol=1e-12)
  out = tn.ncon([a_val, b_val], [[-1, 1, 2], [2, 1, -2]], backend="hip")
  np.testing.assert_allclose(np.asarray(out), np.tensordot(a_val, b_val, [[2, 1], [0, 1]]), rtol=1e-12)
  n = tn.Node(rng.standard_normal((4, 5, 6)), backend="hip")
  l, r, _ = tn.split_node(n, [n[0], n[1]], [n[2]])
  np.testing.assert_allclose(np.asarray((l @ r).tensor), np.asarray(n.tensor), atol=1e-10)


def test_regular_network_sliced_on_gpu(hip):
  """North-star topology, small: sliced contraction on the GPU == greedy on the oracle."""
  from tensornetwork_amd import distributed, workloads as wl
  rng = np.random.default_rng(6)
  tensors = [(rng.standard_normal((4, 4, 4)) * 4 ** -0.75).astype(np.float32) for _ in range(16)]
  ref = float(np.asarray(contractors.greedy(wl.random_regular_network(orc.OracleBackend(), n=16, D=4,
                                                                      tensors=[t.astype(np.float64) for t in tensors])).tensor))
  nodes = wl.random_regular_network(hip, n=16, D=4, tensors=tensors)
  cuts = distributed.choose_cut_edges(nodes, min_slices=8)
  out = float(np.asarray(distributed.contract_sliced(nodes, cuts)))
  assert abs(out - ref) <= 1e-4 * max(abs(ref), 1e-3)
  got = float(np.asarray(contractors.greedy(wl.random_regular_network(hip, n=16, D=4, tensors=tensors)).tensor))
  assert abs(got - ref) <= 1e-4 * max(abs(ref), 1e-3)


def test_mera_layer_on_gpu(hip):
  """Config 5 at chi = 4 (random isometric tensors) vs the oracle, and the D=2 wavelet KAT
  (simple_mera_test.py:48-56: energy -1.242) entirely on the GPU in f64."""
  from tensornetwork_amd import workloads as wl
  ham, rho, iso, dis = wl.mera_random_tensors(4, dtype=np.float32)
  ref = float(np.asarray(wl.mera_energy(orc.OracleBackend(), *(t.astype(np.float64) for t in (ham, rho, iso, dis)),
                                        lambda nodes: contractors.branch(nodes, nbranch=2))))
  dev = [hip.convert_to_tensor(t) for t in (ham, rho, iso, dis)]
  got = float(np.asarray(wl.mera_energy(hip, *dev, lambda nodes: contractors.branch(nodes, nbranch=2))))
  assert abs(got - ref) <= 1e-4 * max(abs(ref), 1.0)

  h = hip.convert_to_tensor(wl.ham_ising())
  w, u = (hip.convert_to_tensor(t) for t in wl.wavelet_mera_tensors())
  s = hip.convert_to_tensor((np.eye(8) / 8).reshape((2,) * 6))
  for _ in range(20):
    s = wl.mera_descend(hip, s, w, u, lambda nodes, order: contractors.greedy(nodes, output_edge_order=order))
  en = float(np.asarray(wl.mera_energy(hip, h, s, w, u, lambda nodes: contractors.branch(nodes, nbranch=2))))
  assert np.isclose(en, -1.242, rtol=1e-3, atol=1e-3)


def test_hipgraph_capture_replay(hip):
  """backend.capture: a chain of small contractions recorded once, replayed on refreshed inputs."""
  rng = np.random.default_rng(11)
  a_host = [rng.standard_normal((12, 12)).astype(np.float32) for _ in range(3)]
  x = hip.convert_to_tensor(a_host[0])
  w1 = hip.convert_to_tensor(rng.standard_normal((12, 12)).astype(np.float32))
  w2 = hip.convert_to_tensor(rng.standard_normal((12, 12)).astype(np.float32))

  def chain(x, w1, w2):
    y = hip.tensordot(x, w1, 1)
    y = hip.transpose(hip.tensordot(y, w2, [[0], [1]]), (1, 0))
    return hip.addition(y, x)

  g = hip.capture(chain, x, w1, w2)
  for host in a_host:
    hip.copy_into(x, hip.convert_to_tensor(host))
    out = np.asarray(g.launch())
    h1, h2 = np.asarray(w1), np.asarray(w2)
    ref = np.tensordot(host @ h1, h2, [[0], [1]]).T + host
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4)
  # blocks of a live graph are not handed to other tensors
  junk = [hip.convert_to_tensor(rng.standard_normal((12, 12)).astype(np.float32)) for _ in range(16)]
  hip.copy_into(x, hip.convert_to_tensor(a_host[1]))
  out = np.asarray(g.launch())
  np.testing.assert_allclose(out, np.tensordot(a_host[1] @ np.asarray(w1), np.asarray(w2), [[0], [1]]).T + a_host[1],
                             rtol=1e-4, atol=1e-4)
  g.close()
  del junk


def test_sliced_contraction_graph_equals_eager(hip):
  from tensornetwork_amd import distributed, workloads as wl
  rng = np.random.default_rng(6)
  tensors = [(rng.standard_normal((4, 4, 4)) * 4 ** -0.75).astype(np.float32) for _ in range(16)]
  nodes = wl.random_regular_network(hip, n=16, D=4, tensors=tensors)
  cuts = distributed.choose_cut_edges(nodes, min_slices=16)
  eager = float(np.asarray(distributed.contract_sliced(nodes, cuts, use_graph=False)))
  graph = float(np.asarray(distributed.contract_sliced(nodes, cuts, use_graph=True)))
  assert abs(eager - graph) <= 1e-5 * max(abs(eager), 1e-3)


def test_k8_rccl_collectives_through_the_c_abi_world1(hip):
  """K8 (include/tnh.h): tnh_comm_unique_id / tnh_comm_init / tnh_allreduce / tnh_allgather /
  tnh_broadcast on the library's own stream, in THIS process (no torch, no second HIP runtime), in a
  1-rank communicator -- RCCL refuses two ranks on one device, so N > 1 is the gloo tests' job
  (same distributed.py code) and the driver's 8-GPU run."""
  import ctypes
  from tensornetwork_amd import comm as tcomm, distributed
  c = tcomm.RcclComm(hip, rank=0, world=1)
  try:
    rank, world = ctypes.c_int(-1), ctypes.c_int(-1)
    _lib.check(hip.lib.tnh_comm_info(ctypes.byref(rank), ctypes.byref(world)))
    assert (rank.value, world.value) == (0, 1)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((5, 7)).astype(np.float32)
    t = hip.convert_to_tensor(x)
    out = c.all_reduce_sum(hip, t)
    assert out.ptr != t.ptr                                    # reduced out of place
    np.testing.assert_array_equal(np.asarray(out), x)
    xb = orc.round_bf16(x)
    out = c.all_reduce_sum(hip, hip.to_bfloat16(xb))           # bf16 partials travel and add as fp32
    assert out.dtype == ta.bfloat16
    np.testing.assert_array_equal(np.asarray(out), xb)
    xc = (x + 2j * x).astype(np.complex64)
    np.testing.assert_array_equal(np.asarray(c.all_reduce_sum(hip, hip.convert_to_tensor(xc))), xc)
    xz = (x + 2j * x).astype(np.complex128)
    np.testing.assert_array_equal(np.asarray(c.all_reduce_sum(hip, hip.convert_to_tensor(xz))), xz)
    xi = rng.integers(-5, 5, size=(3, 4)).astype(np.int64)
    np.testing.assert_array_equal(np.asarray(c.all_reduce_sum(hip, hip.convert_to_tensor(xi))), xi)
    for tt, ref in [(hip.convert_to_tensor(x), x), (hip.to_bfloat16(xb), xb), (hip.convert_to_tensor(xc), xc)]:
      np.testing.assert_array_equal(np.asarray(c.all_gather_rows(hip, tt, [5])), ref)
    assert c.max_over_ranks(3.5) == 3.5 and c.sum_over_ranks(2.0) == 2.0
    c.barrier()
    b = hip.convert_to_tensor(x)
    _lib.check(hip.lib.tnh_broadcast(ctypes.c_void_p(b.ptr), b.nbytes, 0))
    np.testing.assert_array_equal(np.asarray(b), x)
    assert hip.lib.tnh_allreduce(ctypes.c_void_p(b.ptr), b.size, 99, 0) != 0      # bad dtype: status, no abort
    assert hip.lib.tnh_allreduce(ctypes.c_void_p(b.ptr), b.size, _lib.C64, 1) != 0  # complex max: refused
    # the sharded contraction and the sliced network through the same communicator
    a = rng.standard_normal((6, 3, 4)).astype(np.float32)
    bb = rng.standard_normal((4, 3, 2)).astype(np.float32)
    full, bounds = distributed.tensordot_sharded(hip, hip.convert_to_tensor(a), hip.convert_to_tensor(bb),
                                                 [[2, 1], [0, 1]], comm=c)
    assert bounds == (0, 6)
    np.testing.assert_allclose(np.asarray(full), np.tensordot(a, bb, [[2, 1], [0, 1]]), rtol=1e-5, atol=1e-5)
    from tensornetwork_amd import workloads as wl
    tensors = [(rng.standard_normal((4, 4, 4)) * 4 ** -0.75).astype(np.float32) for _ in range(16)]
    nodes = wl.random_regular_network(hip, n=16, D=4, tensors=tensors)
    cuts = distributed.choose_cut_edges(nodes, min_slices=8)
    ref = float(np.asarray(distributed.contract_sliced(nodes, cuts)))
    got = float(np.asarray(distributed.contract_sliced(nodes, cuts, comm=c)))
    assert abs(got - ref) <= 1e-6 * max(abs(ref), 1e-3)
  finally:
    c.close()
  world = ctypes.c_int(-1)
  _lib.check(hip.lib.tnh_comm_info(None, ctypes.byref(world)))
  assert world.value == 0


def test_sliced_bf16_network_accumulates_partials_in_fp32(hip):
  """100+ bf16 slice partials used to be added in bf16.  The sliced bf16 contraction
  must agree with an f32 contraction of the same bf16-rounded tensors to bf16 rounding of the RESULT
  (2^-8 relative), not to sqrt(n_slices) roundings."""
  from tensornetwork_amd import distributed, workloads as wl
  rng = np.random.default_rng(8)
  D, n = 6, 16
  tensors = [orc.round_bf16((rng.standard_normal((D, D, D)) * D ** -0.75)).astype(np.float32) for _ in range(n)]
  nodes32 = wl.random_regular_network(hip, n=n, D=D, tensors=tensors)
  ref = float(np.asarray(contractors.greedy(nodes32).tensor))
  nodes16 = wl.random_regular_network(hip, n=n, D=D, tensors=[hip.to_bfloat16(t) for t in tensors])
  cuts = distributed.choose_cut_edges(nodes16, min_slices=30)
  for use_graph in (False, True):
    out = distributed.contract_sliced(nodes16, cuts, use_graph=use_graph)
    assert out.dtype == ta.bfloat16
    got = float(np.asarray(out))
    # per-slice intermediates are bf16 (as an unsliced bf16 contraction's are); the SUM is not
    assert abs(got - ref) <= 2.0 ** -6 * abs(ref) + 1e-4, (got, ref, use_graph)


def test_sliced_network_keeps_fp32_when_the_backend_contracts_half_into_fp32():
  """with HipBackend(half_output="float32") the per-slice results are fp32 by contract; the sliced sum
  must come back fp32 like the unsliced contraction of the same network (it used to be narrowed to bf16)."""
  from tensornetwork_amd import distributed, workloads as wl
  from tensornetwork_amd.hip_backend import HipBackend
  be = HipBackend(half_output="float32")
  rng = np.random.default_rng(9)
  D, n = 6, 16
  tensors = [orc.round_bf16((rng.standard_normal((D, D, D)) * D ** -0.75)).astype(np.float32) for _ in range(n)]
  nodes = wl.random_regular_network(be, n=n, D=D, tensors=[be.to_bfloat16(t) for t in tensors])
  ref = contractors.greedy(wl.random_regular_network(be, n=n, D=D, tensors=[be.to_bfloat16(t) for t in tensors])).tensor
  cuts = distributed.choose_cut_edges(nodes, min_slices=30)
  for use_graph in (False, True):
    out = distributed.contract_sliced(nodes, cuts, use_graph=use_graph)
    assert out.dtype == ref.dtype == np.float32, (out.dtype, ref.dtype)
    assert abs(float(np.asarray(out)) - float(np.asarray(ref))) <= 2.0 ** -6 * abs(float(np.asarray(ref))) + 1e-4


def test_json_network_from_reference_into_hbm(hip):
  """A network serialised by the reference (NumPy backend, mixed f64 / f32 / complex128 tensors) is
  loaded straight into HBM, contracted on the GPU and written back in the same wire format."""
  import json
  import os
  here = os.path.dirname(os.path.abspath(__file__))
  with open(os.path.join(here, "golden", "network_ref.json")) as f:
    fx = json.load(f)
  ref = np.array(fx["result_re"]) + 1j * np.array(fx["result_im"])
  nodes, binding = network.nodes_from_json(fx["network"], backend=hip)
  assert all(isinstance(n.tensor, ta.DeviceTensor) for n in nodes)
  text = network.nodes_to_json(nodes, edge_binding={k: list(v) for k, v in binding.items()})
  out = contractors.greedy(nodes, output_edge_order=[binding["open"][0], nodes[2][1]]).tensor
  np.testing.assert_allclose(np.asarray(out), ref, rtol=1e-5, atol=1e-5)
  back, _ = network.nodes_from_json(text, backend=orc.OracleBackend())
  for a, b in zip(back, network.nodes_from_json(fx["network"], backend=orc.OracleBackend())[0]):
    np.testing.assert_array_equal(a.tensor, b.tensor)      # serialisation is bit-exact


def test_large_blocks_of_dead_nodes_return_to_the_pool(hip):
  """A Node and its Edges are a reference cycle: `del node` alone does not free the tensor.  A large request that
  the pool cannot serve runs a full collector pass first (device_tensor._Block), so a contract-and-drop loop reuses
  its blocks instead of growing by a hipMalloc per step -- at once with the slack switched off (rounds 1-4: ONE
  block), after a bounded growth with the round-5 policy (passes amortised: up to 2 GiB per request size may be
  granted before the first pass, then the footprint stands still)."""
  import ctypes
  import gc
  from tensornetwork_amd import _lib

  def footprint():
    in_use, cached, peak = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    _lib.check(hip.lib.tnh_mem_stats(ctypes.byref(in_use), ctypes.byref(cached), ctypes.byref(peak)))
    return in_use.value + cached.value

  from tensornetwork_amd import device_tensor as dt
  saved = dict(dt._GC_POLICY)
  per_size = dt._SLACK_PER_SIZE
  ta.configure_gc(freeze=True)     # round 3: the frozen-baseline policy is opt-in (bench.py opts in the same way)
  dt.freeze_collector_baseline()
  gc.collect()
  gc.disable()                     # only the allocator's own collection may run
  try:
    dt._SLACK_PER_SIZE = 0         # no slack: one pass per pool miss
    sizes = []
    for step in range(5):
      node = ta.Node(hip.zeros((24 << 20,), dtype=np.float32), backend=hip)   # 96 MiB, unique size
      assert node[0].node1 is node
      del node
      sizes.append(footprint())
    assert sizes[-1] - sizes[1] == 0, sizes     # step 0 may allocate; afterwards the same block is reused
    dt._SLACK_PER_SIZE = per_size  # the default: growth first, bounded, then one pass per ~21 steps
    gc.collect()
    ta.trim_pool()
    stats0 = ta.gc_stats()
    sizes = []
    for step in range(60):
      node = ta.Node(hip.zeros((26 << 20,), dtype=np.float32), backend=hip)   # 104 MiB, another unique size
      del node
      sizes.append(footprint())
    stats1 = ta.gc_stats()
    assert max(sizes) - sizes[0] <= per_size, (sizes[0], max(sizes))
    assert len(set(sizes[-30:])) == 1, sizes[-30:]                            # stationary once the slack is used up
    skipped = stats1["passes_skipped_for_slack"] - stats0["passes_skipped_for_slack"]
    passes = stats1["full_passes"] - stats0["full_passes"]
    assert skipped == per_size // (104 << 20) and 1 <= passes <= 4, (skipped, passes)
    has = ctypes.c_int(-1)
    _lib.check(hip.lib.tnh_pool_has(1 << 40, ctypes.byref(has)))
    assert has.value == 0
    ta.configure_gc(freeze=saved["freeze"], collect_before_large_alloc=saved["collect"])
  finally:
    dt._SLACK_PER_SIZE = per_size
    gc.enable()
    gc.collect()
    ta.trim_pool()


# ---- round 5: the scheduling modes of contract_sliced and the sliced MERA layer, on the GPU -------------------------
@pytest.mark.parametrize("case", ["16 nodes D=4, two cuts", "16 nodes D=3, three cuts", "open legs, bf16"])
def test_contract_sliced_reuse_equals_slice_by_slice_on_gpu(hip, case):
  """`contract_sliced(reuse=True)` (every step once per distinct value of the cuts it depends on) against
  `reuse=False` (every step in every slice): the SAME slice partials one by one -- same steps, same order, same
  operands, so the f32 partials agree to rounding of the reordered f32 additions only and the bf16 ones exactly --
  and both against the oracle's greedy contraction of the unsliced network."""
  from tensornetwork_amd import distributed, workloads as wl
  rng = np.random.default_rng(11)
  if case.startswith("open"):
    # a ring of 6 rank-3 bf16 tensors with the third legs open: the result is a rank-6 tensor
    raw = [orc.round_bf16(rng.standard_normal((6, 6, 4)) * 6 ** -0.5) for _ in range(6)]
    def build(be, conv):
      nodes = [network.Node(conv(t), backend=be) for t in raw]
      for k in range(6):
        network.connect(nodes[k][1], nodes[(k + 1) % 6][0])
      return nodes, [n[2] for n in nodes]
    nodes, order = build(hip, hip.to_bfloat16)
    cuts = [nodes[0][1], nodes[3][1]]
    ref_nodes, ref_order = build(orc.OracleBackend(), lambda t: t.astype(np.float64))
    ref = np.asarray(contractors.greedy(ref_nodes, output_edge_order=ref_order).tensor)
    tol, exact = 3e-2, True
  else:
    n, D, min_slices = (16, 4, 16) if "D=4" in case else (16, 3, 27)
    tensors = [(rng.standard_normal((D, D, D)) * D ** -0.75).astype(np.float32) for _ in range(n)]
    ref = np.asarray(contractors.greedy(wl.random_regular_network(orc.OracleBackend(), n=n, D=D,
                                                                  tensors=[t.astype(np.float64) for t in tensors])).tensor)
    nodes = wl.random_regular_network(hip, n=n, D=D, tensors=tensors)
    cuts = distributed.choose_cut_edges(nodes, min_slices=min_slices)
    order = None
    assert len(cuts) == (2 if "two" in case else 3)
    tol, exact = 2e-4, False
  p_reuse, p_alone, s_reuse, s_alone = [], [], {}, {}
  out_r = np.asarray(distributed.contract_sliced(nodes, cuts, output_edge_order=order, reuse=True, partials_out=p_reuse,
                                                 stats=s_reuse), dtype=np.float64)
  out_a = np.asarray(distributed.contract_sliced(nodes, cuts, output_edge_order=order, reuse=False, partials_out=p_alone,
                                                 stats=s_alone), dtype=np.float64)
  assert s_reuse["mode"] == "staged" and s_alone["mode"] == "slice by slice"
  assert len(p_reuse) == len(p_alone) == int(np.prod([e.dimension for e in cuts]))
  # the staged run visits the slices in the order of its loop nest; compare as multisets keyed by value
  key = lambda arr: tuple(np.round(np.asarray(arr, dtype=np.float64).reshape(-1)[:4], 10))
  a_sorted, r_sorted = sorted(p_alone, key=key), sorted(p_reuse, key=key)
  scale = max(float(np.max(np.abs(x))) for x in p_alone) + 1e-30
  for x, y in zip(a_sorted, r_sorted):
    if exact:
      np.testing.assert_array_equal(x, y)
    else:
      np.testing.assert_allclose(x, y, rtol=0, atol=2e-6 * scale)
  assert s_reuse["executed_macs"] <= s_alone["executed_macs"]
  np.testing.assert_allclose(out_r, out_a, rtol=tol, atol=tol * float(np.max(np.abs(ref))))
  np.testing.assert_allclose(out_r.reshape(ref.shape), ref, rtol=tol, atol=tol * float(np.max(np.abs(ref))))


def test_contract_sliced_default_mode_is_the_same_on_every_rank_on_gpu(hip):
  """with `reuse=None` every rank must take the same decision, whatever its own block looks like --
  here 9 slices on 4, 5, 7 and 8 emulated ranks (short and empty last blocks), each rank's share contracted on this
  GPU and the shares added: the sum is the one-rank result."""
  from tensornetwork_amd import distributed, workloads as wl
  rng = np.random.default_rng(3)
  tensors = [(rng.standard_normal((3, 3, 3)) * 3 ** -0.75).astype(np.float32) for _ in range(12)]
  nodes = wl.random_regular_network(hip, n=12, D=3, tensors=tensors)
  cuts = distributed.choose_cut_edges(nodes, min_slices=9)
  one = float(np.asarray(distributed.contract_sliced(nodes, cuts)))

  class Rank(distributed.LocalComm):
    def __init__(self, rank, world):
      self.rank, self.world = rank, world

  for world in (4, 5, 7, 8):
    modes, total = set(), 0.0
    for r in range(world):
      st = {}
      total += float(np.asarray(distributed.contract_sliced(nodes, cuts, comm=Rank(r, world), stats=st)))
      modes.add(st["mode"])
    assert len(modes) == 1, (world, modes)
    assert abs(total - one) <= 1e-4 * max(abs(one), 1e-3), (world, total, one)


@pytest.mark.parametrize("placement", ["left", "right"])
def test_mera_sliced_layer_on_gpu(hip, placement):
  """configs[4] semantics at chi = 8: the 64 slices `slice_edge(cut_h, i); slice_edge(cut_rho, j)` leave of ONE layer
  (both nodes of each cut edge sliced: reference network_components.py:1670-1680), contracted with reuse and slice by
  slice on the GPU, against the DENSE layer energy of the same (materialised) tensors on the oracle backend."""
  from tensornetwork_amd import workloads as wl
  chi = 8
  layer = wl.MeraSlicedLayer(hip, chi, placement, np.float32)
  ham, rho, iso, dis = layer.host_tensors()
  dense = float(np.asarray(contractors.branch(wl.mera_layer_network(orc.OracleBackend(), ham, rho, iso, dis, placement),
                                              nbranch=2).tensor))
  staged = wl.mera_sliced_run(hip, chi, placement, np.float32, check_every=16)
  alone = wl.mera_sliced_run(hip, chi, placement, np.float32, reuse_partials=False)
  assert staged["slices_done"] == alone["slices_done"] == chi * chi
  assert abs(staged["energy_partial_sum"] - dense) <= 1e-4 * max(1.0, abs(dense))
  assert abs(alone["energy_partial_sum"] - dense) <= 1e-4 * max(1.0, abs(dense))
  assert staged["stage_runs"] == {"none": 1, "i": chi, "j": chi, "ij": chi * chi}
  assert staged["executed_macs"] == staged["model_macs_with_reuse_all_slices"] < alone["executed_macs"]
  assert staged["checks"] and all(abs(c[1] - c[2]) <= 1e-5 * max(1.0, abs(c[2])) for c in staged["checks"])
  # bf16: reuse and slice-by-slice run the same kernels on the same operands
  h16 = wl.mera_sliced_run(hip, chi, placement, ta.bfloat16)
  a16 = wl.mera_sliced_run(hip, chi, placement, ta.bfloat16, reuse_partials=False)
  assert h16["energy_partial_sum"] == a16["energy_partial_sum"]
