"""The host layers (ncon / network / contractors) driven by the CPU oracle backend
must reproduce the reference's outputs on the golden cases.  The very same
drivers (tests/cases.py) run on the hip backend in the GPU suite."""
import os

import numpy as np
import pytest

import tensornetwork_amd as ta
from oracle.numpy_oracle import OracleBackend
import cases as C


@pytest.fixture(scope="module", params=["oracle", "hip-emulated"])
def be(request):
  """The host layers under both backends: the oracle (NumPy restatement of the reference's backend) and HipBackend on
  the emulated C ABI (tests/emu_tnh.py) -- the second runs the product's own backend code, planning hooks included."""
  if request.param == "oracle":
    yield OracleBackend()
    return
  from emu_tnh import emulated_backend  # pylint: disable=import-outside-toplevel
  with emulated_backend() as hip:
    yield hip


def test_ncon_cases(be, golden):
  for case in golden.cases["ncon"]:
    C.assert_close(C.run_ncon(be, golden, case), golden[case["out"]])


def test_contract_between_cases(be, golden):
  for case in golden.cases["contract_between"]:
    C.assert_close(C.run_contract_between(be, golden, case), golden[case["out"]])


def test_split_node_cases(be, golden):
  for case in golden.cases["split_node"]:
    left, right, trun, recon = C.run_split(be, golden, case)
    assert list(left.shape) == case["left_shape"]
    assert list(right.shape) == case["right_shape"]
    scale = float(np.max(np.abs(golden[case["x"]]))) + 1e-30
    C.assert_close(trun, golden[case["trun"]], scale=scale)
    C.assert_close(recon, golden[case["recon"]], scale=scale)


def test_contractor_cases(be, golden):
  for case in golden.cases["contractors"]:
    C.assert_close(C.run_contractor(be, golden, case), golden[case["out"]])


def test_ncon_readme_example(be):
  # BASELINE config 1: ncon([a, b], [(-1, 1), (1, -2)]) on ones(10, 10)
  a = np.ones((10, 10))
  out = ta.ncon([a, a], [(-1, 1), (1, -2)], backend=be)
  np.testing.assert_array_equal(out, 10 * np.ones((10, 10)))


def test_ncon_errors(be):
  a = np.ones((2, 3))
  with pytest.raises(ValueError):
    ta.ncon([a, a], [[-1, 1]], backend=be)
  with pytest.raises(ValueError):
    ta.ncon([a, a], [[-1, 1], [1, -2]], backend=be)  # dimension mismatch 3 vs 2
  with pytest.raises(ValueError):
    ta.ncon([a], [[-1, -2, -3]], backend=be)


def test_ncon_invalid_network_messages(be):
  """ncon_interface_test.py:630-765: every malformed call and the message the reference's tests match on."""
  a, b, c = np.ones((2, 2)), np.ones((2, 3, 4, 2)), np.ones((2, 4, 4, 3))
  cases_ = [
      (([a, a], [('megan!', 'henry@'), ('henry@', 'megan!')]), {},
       r"only alphanumeric values allowed for string labels, found \['henry@', 'megan!'\]"),
      (([a, a], [(1, 2), (2, 1), (1, 2)]), {}, "number of tensors does not match the number of network connections."),
      (([a, a], [(1,), (1, 2)]), {}, "number of indices does not match number of labels on tensor 0."),
      (([a, a], [(0, 1), (1, 0)]), {}, "only nonzero values are allowed to specify network structure."),
      (([a, a], [(1, 2), (2, 1)]), {"con_order": [-1, 2]},
       r"all number type labels in `con_order` have to be positive, found \[-1\]"),
      (([a, a], [(1, 2), (2, 1)]), {"con_order": ['-hi', 2]},
       r"all string type labels in `con_order` must be unhyphenized, found \['-hi'\]"),
      (([a, a], [(1, 2), (2, 1)]), {"con_order": ['hi', 'hi', 1, 1]},
       r"labels \['hi', 1\] appear more than once in `con_order`."),
      (([a, a], [(1, 2), (2, 1)]), {"con_order": [3, 4, 5]},
       r"`con_order = \[3, 4, 5\] is not a valid contraction order for contracted labels \[1, 2\]"),
      (([a, a], [(1, 2), (2, 1)]), {"con_order": [3, 4]},
       r"labels \[3, 4\] in `con_order` do not appear as contracted labels in `network_structure`."),
      (([a, a], [(-1, 1), (1, -2)]), {"out_order": [-1, 2]},
       r"all number type labels in `out_order` have to be negative, found \[2\]"),
      (([a, a], [('-hi', 1), (1, -2)]), {"out_order": ['hi', -2]},
       r"all string type labels in `out_order` have to be hyphenized, found \['hi'\]"),
      (([a, a], [(1, 2), (2, 1)]), {"out_order": ['-hi', '-hi', -1, -1]},
       r"labels \['-hi', -1\] appear more than once in `out_order`."),
      (([a, a], [(-1, 2), (2, -2)]), {"out_order": [-1, -2, -3]},
       r"`out_order` = \[-1, -2, -3\] is not a valid output order for open labels \[-1, -2\]"),
      (([a, a], [(-1, 2), (2, -2)]), {"out_order": [-3, -4]},
       r"labels \[-3, -4\] in `out_order` do not appear in `network_structure`."),
      (([b, c], [(1, 2, 3, 4), (1, 2, 3, 4)]), {}, r"tensor dimensions for labels \[2, 4\] are mismatching"),
  ]
  for args, kwargs, message in cases_:
    with pytest.raises(ValueError, match=message):
      ta.ncon(*args, backend=be, **kwargs)
    wrapped = [ta.Tensor(t, backend=be) for t in args[0]]          # test_node_invalid_network: Tensor operands
    with pytest.raises(ValueError, match=message):
      ta.ncon(wrapped, args[1], backend=be, **kwargs)
  ones3, ones2 = np.ones((2, 2, 2)), np.ones((2, 2))
  with pytest.raises(ValueError, match=r"ncon seems stuck in an infinite loop. \nPlease check if `con_order` = \[3\] is a "
                     r"valid contraction order for \n`network_structure` = \[\[3, 1, 2\], \[3, 1\], \[3, 2\]\]"):
    ta.ncon([ones3, ones2, ones2], [[3, 1, 2], [3, 1], [3, 2]], con_order=[3], check_network=False, backend=be)


def test_einsum_matches_numpy(be):
  rng = np.random.default_rng(3)
  a, b, c = rng.standard_normal((3, 4)), rng.standard_normal((4, 5)), rng.standard_normal((5, 3))
  for expr, ops in [("ij,jk->ik", (a, b)), ("ij,jk,ki->", (a, b, c)), ("ij,jk", (a, b)),
                    ("ii->", (rng.standard_normal((4, 4)),)), ("ij->j", (a,)), ("bij,bjk->bik",
                     (rng.standard_normal((2, 3, 4)), rng.standard_normal((2, 4, 5))))]:
    np.testing.assert_allclose(ta.einsum(expr, *ops, backend=be), np.einsum(expr, *ops), rtol=1e-12,
                               atol=1e-12)


def test_node_edge_bookkeeping(be):
  a = ta.Node(np.arange(24.0).reshape(2, 3, 4), backend=be)
  b = ta.Node(np.arange(12.0).reshape(4, 3), backend=be)
  e = a[2] ^ b[0]
  assert not e.is_dangling() and e.dimension == 4
  with pytest.raises(ValueError):
    ta.connect(a[2], b[1])  # already connected
  with pytest.raises(ValueError):
    ta.connect(a[0], b[1])  # dimension mismatch
  c = a @ b
  assert c.shape == (2, 3, 3)
  np.testing.assert_allclose(c.tensor, np.tensordot(a.tensor, b.tensor, [[2], [0]]))
  c.reorder_edges([c[2], c[0], c[1]])
  assert c.shape == (3, 2, 3)
  with pytest.raises(ValueError):
    ta.contract_between(c, ta.Node(np.ones(2), backend=be))  # no shared edge


def test_slice_and_copy(be):
  rng = np.random.default_rng(5)
  a = ta.Node(rng.standard_normal((3, 4)), backend=be)
  b = ta.Node(rng.standard_normal((4, 5)), backend=be)
  e = a[1] ^ b[0]
  full = (a.tensor @ b.tensor)
  total = 0
  for i in range(4):
    node_map, edge_map = ta.copy([a, b])
    ta.slice_edge(edge_map[e], i, 1)
    total = total + ta.contract_between(node_map[a], node_map[b]).tensor
  np.testing.assert_allclose(total, full, rtol=1e-12)


def test_json_wire_format_interchange_with_reference():
  """a network written by the reference's tn.nodes_to_json (NumPy backend; fixture
  tests/golden/network_ref.json, made by make_golden_json.py) loads through nodes_from_json, contracts to
  the reference's result, and survives a round trip through our own nodes_to_json."""
  import json
  import os
  from tensornetwork_amd import contractors, network
  here = os.path.dirname(os.path.abspath(__file__))
  with open(os.path.join(here, "golden", "network_ref.json")) as f:
    fx = json.load(f)
  be = OracleBackend()
  ref = np.array(fx["result_re"]) + 1j * np.array(fx["result_im"])
  nodes, binding = network.nodes_from_json(fx["network"], backend=be)
  assert [n.name for n in nodes] == ["a", "b", "c"] and nodes[0].axis_names == ["x", "y", "z"]
  assert set(binding) == {"bond", "pair", "open"} and len(binding["pair"]) == 2
  assert binding["bond"][0].name == "ab" and binding["open"][0].is_dangling()
  assert nodes[1].tensor.dtype == np.float32 and np.iscomplexobj(nodes[2].tensor)
  text = network.nodes_to_json(nodes, edge_binding={k: list(v) for k, v in binding.items()})
  out = contractors.greedy(nodes, output_edge_order=[binding["open"][0], nodes[2][1]]).tensor
  np.testing.assert_allclose(out, ref, rtol=1e-6)
  nodes2, binding2 = network.nodes_from_json(text, backend=be)
  # a sub-network: the edge to the excluded node keeps its attributes but loses that end
  text3 = network.nodes_to_json(nodes2[:2])
  part, _ = network.nodes_from_json(text3, backend=be)
  assert part[1][2].is_dangling() and part[1][2].name == "bc"
  out2 = contractors.greedy(nodes2, output_edge_order=[binding2["open"][0], nodes2[2][1]]).tensor
  np.testing.assert_allclose(out2, ref, rtol=1e-6)
  with pytest.raises(TypeError):
    network.nodes_to_json(part, edge_binding={1: part[0][0]})


@pytest.mark.parametrize("seed", range(8))
def test_contractors_with_layout_planning_random_networks(seed):
  """Layout planning (edge contraction times -> free-axis order hints + operand swaps) must never change a
  result: random connected networks with open legs, greedy / optimal / branch, against np.einsum.  The
  oracle backend's tensordot_planned ALWAYS applies the requested orders, the worst case for the edge
  bookkeeping."""
  import string
  from tensornetwork_amd import contractors, network
  rng = np.random.default_rng(100 + seed)
  be = OracleBackend()
  n_nodes = int(rng.integers(3, 7))
  letters = iter(string.ascii_letters)
  labels = [[] for _ in range(n_nodes)]
  dims = {}
  # a random spanning tree plus a few extra bonds keeps the network connected
  pairs = [(i, int(rng.integers(0, i))) for i in range(1, n_nodes)]
  pairs += [tuple(sorted(rng.choice(n_nodes, 2, replace=False).tolist())) for _ in range(int(rng.integers(0, 4)))]
  for a, b in pairs:
    l = next(letters)
    dims[l] = int(rng.integers(2, 5))
    labels[a].append(l)
    labels[b].append(l)
  open_labels = []
  for i in range(n_nodes):
    for _ in range(int(rng.integers(0, 3))):
      l = next(letters)
      dims[l] = int(rng.integers(2, 4))
      labels[i].append(l)
      open_labels.append(l)
  for i in range(n_nodes):
    rng.shuffle(labels[i])
  arrays = [rng.standard_normal([dims[l] for l in labs]) for labs in labels]
  rng.shuffle(open_labels)
  ref = np.einsum(",".join("".join(l) for l in labels) + "->" + "".join(open_labels), *arrays)
  for contractor in (contractors.greedy, contractors.optimal, lambda nd, **kw: contractors.branch(nd, nbranch=2, **kw)):
    nodes = [network.Node(x, backend=be) for x in arrays]
    first = {}
    out_edges = {}
    for i, labs in enumerate(labels):
      for ax, l in enumerate(labs):
        if l in open_labels:
          out_edges[l] = nodes[i][ax]
        elif l in first:
          network.connect(first[l], nodes[i][ax])
        else:
          first[l] = nodes[i][ax]
    res = contractor(nodes, output_edge_order=[out_edges[l] for l in open_labels])
    np.testing.assert_allclose(res.tensor, ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("seed", range(6))
def test_ncon_with_layout_planning_random(seed):
  """ncon's own use of tensordot_planned (labels' contraction order -> hints / operand swaps) vs np.einsum."""
  import string
  rng = np.random.default_rng(200 + seed)
  be = OracleBackend()
  n = int(rng.integers(3, 6))
  struct = [[] for _ in range(n)]
  dims, next_pos, next_neg = {}, 1, -1
  for i in range(1, n):
    j = int(rng.integers(0, i))
    for _ in range(int(rng.integers(1, 3))):
      dims[next_pos] = int(rng.integers(2, 5))
      struct[i].append(next_pos)
      struct[j].append(next_pos)
      next_pos += 1
  for i in range(n):
    for _ in range(int(rng.integers(0, 2))):
      dims[next_neg] = int(rng.integers(2, 4))
      struct[i].append(next_neg)
      next_neg -= 1
  for s in struct:
    rng.shuffle(s)
  arrays = [rng.standard_normal([dims[l] for l in s]) for s in struct]
  names = {l: string.ascii_letters[k] for k, l in enumerate(sorted(dims, key=lambda x: (x < 0, abs(x))))}
  out = "".join(names[l] for l in sorted([l for l in dims if l < 0], reverse=True))
  ref = np.einsum(",".join("".join(names[l] for l in s) for s in struct) + "->" + out, *arrays)
  got = ta.ncon(arrays, struct, backend=be)
  np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


def test_graph_surgery_reference_cases():
  """flatten / split / disconnect / remove / redirect / reduced_density on the oracle backend
  (the GPU suite runs the same checker on HipBackend)."""
  import cases
  cases.check_graph_surgery(OracleBackend(), 1e-12)


def test_host_bf16_rounding_equals_its_one_line_definition():
  """device_tensor.f32_to_bf16_bits works through the array in cache-sized pieces with in-place integer steps (round 5:
  14x faster on 50 M elements); its results are the bits of the one-line definition -- round to nearest even on the
  uint32 image, NaNs keep sign and top payload with the quiet bit set -- for random bit patterns, the special values,
  every shape (0-d, empty, non-contiguous, float64 input) and arrays longer than one piece."""
  from tensornetwork_amd import device_tensor as dt

  def one_line(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    nan = (u & np.uint32(0x7fffffff)) > np.uint32(0x7f800000)
    r = (u + (np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1)))) >> np.uint32(16)
    return np.where(nan, (u >> np.uint32(16)) | np.uint32(0x40), r).astype(np.uint16)

  rng = np.random.default_rng(0)
  bits = rng.integers(0, 2**32, size=(1 << 21) + 12345, dtype=np.uint64).astype(np.uint32)      # more than two pieces
  special = np.array([0, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0x7f800001, 0xffffffff, 0x7f7fffff, 0x7f7f8000,
                      0x7f7f7fff, 0x00000001, 0x00008000, 0x00018000, 0x3f808000, 0x3f818000, 0x3f80ffff, 0xffc00001,
                      0x7fffffff], dtype=np.uint32)
  x = np.concatenate([bits, special]).view(np.float32)
  with np.errstate(all="ignore"):
    want, got = one_line(x), dt.f32_to_bf16_bits(x)
  assert got.dtype == np.uint16 and got.shape == want.shape and np.array_equal(got, want)
  for shape in [(), (1,), (3, 5), (0,), (2, 0, 3)]:
    y = rng.standard_normal(shape).astype(np.float32)
    assert dt.f32_to_bf16_bits(y).shape == one_line(y).shape and np.array_equal(dt.f32_to_bf16_bits(y), one_line(y)), shape
  y = rng.standard_normal((64, 33))[:, ::2]                       # float64, non-contiguous
  assert np.array_equal(dt.f32_to_bf16_bits(y), one_line(y))
  np.testing.assert_array_equal(dt.round_to_bf16(np.array([1.0, 1.00390625, 1.01171875], dtype=np.float32)), [1.0, 1.0, 1.015625])


def test_collector_policy_before_large_allocations():
  """device_tensor: a full collection in front of a pool miss only where it is cheaper than the hipMalloc."""
  from tensornetwork_amd import device_tensor as dt
  saved = dt._gc_cost_seconds
  try:
    dt._gc_cost_seconds = 8e-3                       # an application with a large live heap
    assert not dt._worth_collecting(32 << 20)        # below the floor: never
    assert not dt._worth_collecting(67 << 20)        # hipMalloc ~1.8 ms < 8 ms
    assert dt._gc_cost_seconds < 8e-3                # skipped opportunities decay the estimate
    assert dt._worth_collecting(8 << 30)             # ~224 ms of hipMalloc
    dt._gc_cost_seconds = 0.0                        # frozen baseline, small heap
    assert dt._worth_collecting(64 << 20)
    dt._collect_and_time()
    assert 0.0 < dt._gc_cost_seconds < 5.0
  finally:
    dt._gc_cost_seconds = saved


def test_collector_passes_are_amortised_over_a_bounded_growth_of_the_pool(monkeypatch):
  """device_tensor, round 5: a step whose result dies in a Node <-> Edge cycle needs a FULL collector pass to get
  its block back, and the pass costs what the host's heap costs (measured on the MI355X box: 1.3 ms per step once
  21.7k objects sat behind the freeze -- the D = 96 sweep row host-bound at 1028 instead of 1364 TFLOP/s).  Requests
  of up to 512 MiB may skip the pass and grow the pool instead, 2 GiB per request size at most; then ONE pass returns
  all the dead blocks of the skipped steps.  Fake library with a real size-keyed free list and tnh_mem_stats."""
  import gc
  import tensornetwork_amd as ta
  from tensornetwork_amd import _lib, device_tensor as dt

  class PoolLib:
    def __init__(self):
      self.free, self.live, self.next, self.misses = {}, {}, 4096, 0
    def tnh_pool_has(self, nbytes, has_ref):
      has_ref._obj.value = 1 if self.free.get(int(nbytes)) else 0
      return 0
    def tnh_malloc(self, pref, nbytes):
      lst = self.free.get(int(nbytes))
      if lst:
        p = lst.pop()
      else:
        self.misses += 1
        p, self.next = self.next, self.next + 4096
      self.live[p] = int(nbytes)
      pref._obj.value = p
      return 0
    def tnh_free(self, p):
      p = p.value if hasattr(p, "value") else int(p)
      self.free.setdefault(self.live.pop(p), []).append(p)
      return 0
    def tnh_mem_stats(self, in_use, cached, peak):  # pylint: disable=unused-argument
      in_use._obj.value = sum(self.live.values())
      cached._obj.value = sum(k * len(v) for k, v in self.free.items())
      return 0
    def tnh_trim(self):
      self.free.clear()
      return 0

  class Holder:                 # a Node <-> Edge pair in miniature: a reference cycle that owns a block
    def __init__(self, block):
      self.block, self.me = block, self

  saved = (dict(dt._GC_POLICY), dt._gc_cost_seconds, dict(dt._gc_stats), dict(dt._slack_granted), _lib._lib, _lib._device)
  lib = PoolLib()
  was_enabled = gc.isenabled()
  size = 170 << 20
  try:
    _lib._lib, _lib._device = lib, 0
    gc.collect()
    gc.disable()                # no automatic collection interferes with the counts
    dt._slack_granted.clear()
    for k in dt._gc_stats:
      dt._gc_stats[k] = 0
    dt._gc_cost_seconds = 0.0   # a cheap pass: always "worth it" once the slack is used up ...
    timed = dt._collect_and_time

    def cheap_pass():           # ... whatever a pass over THIS process' heap (pytest's) costs
      timed()
      dt._gc_cost_seconds = 0.0
    monkeypatch.setattr(dt, "_collect_and_time", cheap_pass)
    ta.configure_gc(collect_before_large_alloc=True)
    steps = 60
    for _ in range(steps):
      Holder(dt._Block(size))
    st = ta.gc_stats()
    grants = (2 << 30) // size                                           # 12 blocks of 170 MiB fit into 2 GiB
    assert st["passes_skipped_for_slack"] == grants == lib.misses        # only the granted requests reached "hipMalloc"
    assert st["slack_granted_bytes"] == grants * size
    # from then on one pass per `grants` steps (each returns every dead block), not one per step
    assert 1 <= st["full_passes"] <= (steps - grants) // grants + 1
    # a request above 512 MiB never skips: one pass per miss, as before
    before = ta.gc_stats()
    for _ in range(3):
      Holder(dt._Block(600 << 20))
    after = ta.gc_stats()
    assert after["passes_skipped_for_slack"] == before["passes_skipped_for_slack"]
    assert after["full_passes"] - before["full_passes"] == 3
    # a trim past device_tensor.trim() is noticed (the pool holds less than was granted) and the count starts again
    gc.collect()
    lib.tnh_trim()
    assert ta.gc_stats()["slack_granted_bytes"] == grants * size
    Holder(dt._Block(size))
    assert ta.gc_stats()["slack_granted_bytes"] == size
    ta.trim_pool()
    assert ta.gc_stats()["slack_granted_bytes"] == 0
    # collect_before_large_alloc=False: neither a pass nor a grant
    ta.configure_gc(collect_before_large_alloc=False)
    before = ta.gc_stats()
    for _ in range(3):
      Holder(dt._Block(size))
    after = ta.gc_stats()
    assert (after["full_passes"], after["passes_skipped_for_slack"]) == (before["full_passes"], before["passes_skipped_for_slack"])
  finally:
    if was_enabled:
      gc.enable()
    gc.collect()                 # the fake library takes its own blocks back
    _lib._lib, _lib._device = saved[4], saved[5]
    dt._GC_POLICY.update(saved[0])
    dt._gc_cost_seconds = saved[1]
    dt._gc_stats.update(saved[2])
    dt._slack_granted.clear()
    dt._slack_granted.update(saved[3])


def test_ncon_solver_known_mera_cost_and_consistency():
  """nconinterface_test.py:22-66: the binary-MERA network has optimal cost 2 chi^9 + 4 chi^8 + 2 chi^6 +
  2 chi^5 multiplications; on random networks the reported cost equals the cost of the returned order."""
  from tensornetwork_amd import pathfinder
  for chi in (2, 3, 5):
    u, w, ham = np.zeros((chi,) * 4), np.zeros((chi,) * 3), np.zeros((chi,) * 6)
    tensors = [u, u, w, w, w, ham, u, u, w, w, w]
    connects = [[1, 3, 10, 11], [4, 7, 12, 13], [8, 10, -4], [11, 12, -5], [13, 14, -6], [2, 5, 6, 3, 4, 7],
                [1, 2, 9, 17], [5, 6, 16, 15], [8, 9, -1], [17, 16, -2], [15, 14, -3]]
    con_order, cost, is_optimal = pathfinder.ncon_solver(tensors, connects, max_branch=None)
    assert is_optimal
    np.testing.assert_allclose(cost, np.log10(2 * chi**9 + 4 * chi**8 + 2 * chi**6 + 2 * chi**5))
    assert sorted(con_order) == list(range(1, 18))
    np.testing.assert_allclose(pathfinder.ncon_cost_check(tensors, connects, con_order), cost)
    # the order drives ncon to the same value as the default order
    rng = np.random.default_rng(chi)
    vals = [rng.standard_normal(t.shape) for t in tensors]
    be = OracleBackend()
    np.testing.assert_allclose(ta.ncon(vals, connects, con_order=list(con_order), backend=be),
                               ta.ncon(vals, connects, backend=be), rtol=1e-9)
  rng = np.random.default_rng(0)
  chi, n = 4, 8
  for num_closed in (1, 5, 9, 14):
    num_open = 4 * n - 2 * num_closed
    cl, op = 1 + np.arange(num_closed), -1 - np.arange(num_open)
    comb = np.concatenate((op, cl, cl))[rng.permutation(4 * n)]
    connects = []
    for k in range(n):
      ring = [num_closed + k + 1, num_closed + k + 2 if k < n - 1 else num_closed + 1]
      labs = np.concatenate((comb[4 * k:4 * (k + 1)], ring))
      connects.append([int(x) for x in labs[rng.permutation(6)]])
    tensors = [np.zeros((chi,) * 6)] * n
    for max_branch in (1, 3):
      con_order, cost, _ = pathfinder.ncon_solver(tensors, connects, max_branch=max_branch)
      assert sorted(con_order) == list(range(1, num_closed + n + 1))
      np.testing.assert_allclose(pathfinder.ncon_cost_check(tensors, connects, con_order), cost)


def test_default_backend_stack():
  # backend_contextmanager.py:14-52 / tests/backend_contextmanager_test.py
  be = OracleBackend()
  assert ta.get_default_backend() == "hip"
  with ta.DefaultBackend(be):
    assert ta.Node(np.eye(2)).backend is be
    inner = OracleBackend()
    with ta.DefaultBackend(inner):
      assert ta.Node(np.eye(2)).backend is inner
      with pytest.raises(AssertionError, match="should not be changed inside"):
        ta.set_default_backend(be)
    assert ta.Node(np.eye(2)).backend is be
    np.testing.assert_allclose(ta.ncon([np.eye(2), np.ones(2)], [[-1, 1], [1]]), np.ones(2))
  assert ta.get_default_backend() == "hip"
  ta.set_default_backend(be)
  try:
    assert ta.Node(np.eye(2)).backend is be and ta.Tensor(np.eye(2)).backend is be
  finally:
    ta.set_default_backend("hip")
  with pytest.raises(ValueError):
    ta.set_default_backend(-1)
  with pytest.raises(ValueError, match="was not found"):
    ta.set_default_backend("BAD_NAME")
  with pytest.raises(ValueError):
    ta.DefaultBackend(-1)


@pytest.mark.parametrize("seed", range(150))
def test_ncon_fuzz_against_einsum(seed):
  """Random ncon calls with everything the reference's tests exercise at once (ncon_interface_test.py:
  outer products, partial and batched traces, batched matmuls, hyper-indices, explicit orders):
  ncon == einsum with the negative labels as output and every positive label summed."""
  import string
  rng = np.random.default_rng(1000 + seed)
  be = OracleBackend()
  n_t = int(rng.integers(1, 5))
  n_pos, n_neg = int(rng.integers(0, 5)), int(rng.integers(0, 4))
  labels = [int(x) for x in range(1, n_pos + 1)] + [-int(x) for x in range(1, n_neg + 1)]
  dims = {l: int(rng.integers(2, 4)) for l in labels}
  structure = [[] for _ in range(n_t)]
  for l in labels:
    copies = int(rng.integers(2, 4)) if l > 0 else int(rng.integers(1, 3))
    for _ in range(copies):
      structure[int(rng.integers(0, n_t))].append(l)
  for s in structure:
    # at most two copies of a label per tensor (a trace / a batched diagonal), ranks <= 6
    for l in set(s):
      while s.count(l) > 2:
        s.remove(l)
    del s[6:]
    rng.shuffle(s)
  used = [l for s in structure for l in s]
  pos = sorted({l for l in used if l > 0})
  neg = sorted({l for l in used if l < 0}, reverse=True)
  if any(used.count(l) == 1 for l in pos):
    pytest.skip("a positive label on a single axis is a plain sum: the reference rejects it")
  if any(s.count(l) == 2 for s in structure for l in neg):
    pytest.skip("repeated open label on one tensor")
  tensors = [rng.standard_normal([dims[l] for l in s]) for s in structure]
  letter = {l: string.ascii_letters[k] for k, l in enumerate(pos + neg)}
  out_order = list(rng.permutation(neg)) if neg and rng.random() < 0.5 else None
  con_order = [int(x) for x in rng.permutation(pos)] if pos and rng.random() < 0.5 else None
  final = [int(x) for x in out_order] if out_order is not None else neg
  expr = ",".join("".join(letter[l] for l in s) for s in structure) + "->" + "".join(letter[l] for l in final)
  want = np.einsum(expr, *tensors)
  got = ta.ncon(tensors, structure, con_order=con_order, out_order=None if out_order is None else final, backend=be)
  np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)


def test_node_and_edge_accessors(be):
  """network_components_free_test.py: the small Node / Edge API (axis names, add_edge, get_dimension, slices,
  ordering, copy, update_axis, is_being_used, `edge | edge`)."""
  n = ta.Node(np.arange(24.0).reshape(2, 3, 4), name="n", axis_names=["a", "b", "c"], backend=be)
  assert n.get_rank() == 3 and n.get_axis_number("b") == 1 and n.get_axis_number(2) == 2
  assert n.get_dimension("c") == 4 and n.get_dimension(0) == 2 and n["b"] is n[1] and n[0:2] == n.edges[:2]
  assert n.get_all_edges() == n.edges and n.get_all_edges() is not n.edges
  assert str(n) == "n" and n.has_dangling_edge() and not n.has_nondangling_edge() and n.sparse_shape == (2, 3, 4)
  with pytest.raises(ValueError, match="Axis name 'zz' not found"):
    n.get_axis_number("zz")
  with pytest.raises(ValueError, match="Axis must be positive and less than rank"):
    n.get_dimension(5)
  n.add_axis_names(["x", "y", "z"])
  assert n.axis_names == ["x", "y", "z"]
  with pytest.raises(ValueError, match="Not all axis names are unique"):
    n.add_axis_names(["x", "x", "z"])
  with pytest.raises(ValueError, match="axis_names is not the same length"):
    n.add_axis_names(["x"])
  with pytest.raises(TypeError, match="axis_names should be str type"):
    n.add_axis_names(["x", "y", 3])
  n.set_name("m")
  assert n.name == "m"
  with pytest.raises(TypeError):
    n.set_name(3)
  other = ta.Node(np.ones((4, 2)), backend=be)
  e = ta.connect(n[2], other[0], name="bond")
  assert n.has_nondangling_edge() and e.is_being_used() and str(e) == "bond" and sorted([n, other])[0] in (n, other)
  with pytest.raises(ValueError, match="is not a Node type"):
    n < 3  # pylint: disable=pointless-statement
  with pytest.raises(TypeError):
    e < 3  # pylint: disable=pointless-statement
  with pytest.raises(TypeError, match="Cannot use '@' with type"):
    n @ 3  # pylint: disable=pointless-statement
  with pytest.raises(ValueError, match="already has a non-dangling edge"):
    n.add_edge(ta.Edge(n, 2), 2)
  with pytest.raises(ValueError, match="Axis must be positive"):
    n.add_edge(ta.Edge(n, 0), 7)
  fresh = ta.Edge(n, 0, name="fresh")
  n.add_edge(fresh, "x")
  assert n[0] is fresh
  e.set_name("renamed")
  assert e.name == "renamed"
  with pytest.raises(TypeError):
    e.set_name(1)
  with pytest.raises(ValueError, match="Cannot break two unconnected edges"):
    e | fresh  # pylint: disable=pointless-statement
  left, right = e | e
  assert left.is_dangling() and right.is_dangling() and n[2] is left and other[0] is right and not e.is_being_used()
  with pytest.raises(ValueError, match="did not contain node"):
    left.update_axis(0, other, 1, other)
  left.update_axis(2, n, 2, n)
  # copy: trace edges survive, everything else dangles; conjugation
  z = ta.Node(np.arange(8.0).reshape(2, 2, 2) * (1 + 1j), name="z", backend=be)
  tr = ta.connect(z[0], z[2], name="loop")
  ta.connect(z[1], ta.Node(np.ones(2), backend=be)[0])
  zc = z.copy(conjugate=True)
  assert zc is not z and zc.name == "z" and zc[0] is zc[2] and zc[0].is_trace() and zc[0].name == "loop" and zc[1].is_dangling()
  np.testing.assert_allclose(np.asarray(zc.tensor), np.conj(np.asarray(z.tensor)))
  assert tr.is_trace()


def test_jit_decorator(be):
  # backends/decorators_test.py: fixed backend, backend from an argument, error cases
  def fun(x, backend_arg, y):
    return x * 2 + y
  assert ta.jit(fun, backend=be)(1, None, 3) == 5
  by_arg = ta.jit(fun, backend_argnum=1)
  assert by_arg(1, be, 3) == 5 and by_arg.__name__ == "fun"
  with pytest.raises(ValueError, match="backend must be None if backend_argnum is specified"):
    ta.jit(fun, backend=be, backend_argnum=1)
  with pytest.raises(ValueError, match="did not specify a backend"):
    by_arg(1, "BAD_NAME", 3)
  with pytest.raises(ValueError, match="did not specify a backend"):
    by_arg(1, 7, 3)


def test_node_collection(be):
  # network_components_free_test.py:1232-1275
  box = []
  with ta.NodeCollection(box):
    a = ta.Node(np.eye(2), backend=be)
    b = ta.Node(np.eye(3), backend=be)
  c = ta.Node(np.eye(2), backend=be)
  assert box == [a, b] and c not in box
  bag = set()
  with ta.NodeCollection(bag):
    a = ta.CopyNode(rank=4, dimension=3, name="copier1", backend=be)
    b = ta.Node(np.eye(3), backend=be)
  assert bag == {a, b}
  outer, inner = set(), set()
  with ta.NodeCollection(outer):
    with ta.NodeCollection(inner):
      a, b = ta.Node(np.eye(2), backend=be), ta.Node(np.eye(3), backend=be)
    d = ta.Node(np.eye(2), backend=be)
  assert inner == {a, b} and outer == {d}
  with pytest.raises(ValueError, match="must be list or set"):
    ta.NodeCollection({})


def test_node_name_type_checks(be):
  # network_components_free_test.py:1290-1330
  with pytest.raises(TypeError, match="Node name should be str type"):
    ta.Node(np.eye(2), name=["A"], backend=be)
  with pytest.raises(TypeError, match="Node name should be str type"):
    ta.Node(np.eye(2), name=1, backend=be)
  with pytest.raises(TypeError, match="axis_names should be str type"):
    ta.Node(np.eye(2), axis_names=[0, 1], backend=be)
  with pytest.raises(ValueError, match="axis_names is not the same length"):
    ta.Node(np.eye(2), axis_names=["a"], backend=be)


def test_operand_view_addressing_reproduces_tensordot():
  """Host side of the transpose-absorbing lowering (hip_backend._operand_view -> tnh_gemm_view): whenever a
  view is offered, the two-level address formula of include/tnh.h must enumerate exactly the matrix
  that transpose + reshape would have materialised (checked by evaluating the formula in NumPy)."""
  import itertools
  from tensornetwork_amd.hip_backend import _operand_view

  def matrix_from_view(flat, v, rows, K):
    r = np.arange(rows)[:, None]
    k = np.arange(K)[None, :]
    idx = (r // v.r0) * v.sr1 + (r % v.r0) * v.sr0 + (k // v.k0) * v.sk1 + (k % v.k0) * v.sk0
    return flat[idx]

  rng = np.random.default_rng(0)
  offered = 0
  for shape_a, shape_b, axes in [
      ((8, 64, 8, 64), (64, 8, 64, 16), ([1, 3], [2, 0])),      # config-2 layout L1: zero permutes
      ((8, 8, 64, 64), (64, 64, 8, 16), ([2, 3], [0, 1])),      # layout L0: a K-contiguous, b k-major
      ((64, 64, 8, 8), (64, 64, 16, 8), ([0, 1], [0, 1])),      # both k-major
      ((8, 128), (16, 128), ([1], [1])),                         # plain NT
      ((128, 8), (128, 16), ([0], [0])),                         # plain TN
      ((2, 8, 64, 1, 64), (64, 64, 8, 3), ([2, 4], [0, 1])),    # size-1 axis in between
      ((8, 64, 8, 64), (64, 8, 64, 16), ([3, 1], [0, 2])),      # same pairs listed in the other order
      ((8, 96, 8, 96), (96, 8, 96, 16), ([1, 3], [2, 0])),      # D = 96: runs of 96 = three half K-tiles
      ((8, 32, 8, 32), (32, 32, 8, 16), ([1, 3], [0, 1])),      # chi = 32: runs of 32
  ]:
    a = rng.standard_normal(shape_a)
    b = rng.standard_normal(shape_b)
    axes_a, axes_b = axes
    free_a = [i for i in range(a.ndim) if i not in axes_a]
    free_b = [i for i in range(b.ndim) if i not in axes_b]
    ref = np.tensordot(a, b, axes)
    nc = len(axes_a)
    for order in (sorted(range(nc), key=lambda i: axes_a[i]), sorted(range(nc), key=lambda i: axes_b[i])):
      va = _operand_view(a.shape, free_a, [axes_a[i] for i in order])
      vb = _operand_view(b.shape, free_b, [axes_b[i] for i in order])
      if va is None or vb is None:
        continue
      offered += 1
      m = int(np.prod([a.shape[i] for i in free_a]))
      n = int(np.prod([b.shape[i] for i in free_b]))
      k = int(np.prod([a.shape[i] for i in axes_a]))
      am = matrix_from_view(a.reshape(-1), va, m, k)
      bm = matrix_from_view(b.reshape(-1), vb, n, k)
      np.testing.assert_allclose((am @ bm.T).reshape(ref.shape), ref, rtol=1e-12, atol=1e-12)
      assert (va.sk0 == 1) != (va.sr0 == 1) and (vb.sk0 == 1) != (vb.sr0 == 1)
      break
    else:
      raise AssertionError(f"no in-place view offered for {shape_a} x {shape_b} {axes}")
  assert offered == 9
  # refused: inner contraction run not a multiple of 32, three memory runs, misaligned strides
  assert _operand_view((8, 48, 8, 48), [0, 2], [1, 3]) is None
  assert _operand_view((4, 64, 4, 64, 4, 64), [0, 2, 4], [1, 3, 5]) is None
  assert _operand_view((8, 64, 8, 68), [0, 1, 2], [3]) is None       # contraction run 68
  assert _operand_view((64, 12), [1], [0]) is None                   # k-major with 12-element rows (not 16-B chunks)


def test_gc_policy_opt_out_leaves_the_collector_alone(monkeypatch):
  """a library must not freeze / run the host's collector without an opt-out that is
  part of the API.  With configure_gc(False, False) neither gc.freeze nor gc.collect is reached."""
  import gc
  from tensornetwork_amd import device_tensor as dt
  import tensornetwork_amd as ta
  saved = (dict(dt._GC_POLICY), dt._GC_FROZEN, dt._gc_cost_seconds)
  calls = []
  monkeypatch.setattr(gc, "freeze", lambda: calls.append("freeze"))
  monkeypatch.setattr(gc, "collect", lambda *a: calls.append("collect") or 0)
  monkeypatch.setattr(gc, "unfreeze", lambda: calls.append("unfreeze"))
  monkeypatch.delenv("TNH_GC_FREEZE", raising=False)
  try:
    dt._GC_FROZEN = False
    # round 3: freezing the host's heap is opt-in -- the default policy never reaches gc.freeze()
    dt._GC_POLICY.update({"freeze": False, "collect": True})
    dt.freeze_collector_baseline()
    assert calls == [] and not dt._GC_FROZEN
    assert ta.configure_gc(freeze=False, collect_before_large_alloc=False) == {"freeze": False, "collect": False}
    dt.freeze_collector_baseline()
    assert not dt._worth_collecting(64 << 30)
    assert calls == []
    assert ta.configure_gc(freeze=True, collect_before_large_alloc=True) == {"freeze": True, "collect": True}
    dt.freeze_collector_baseline()
    assert calls == ["collect", "freeze"] and dt._GC_FROZEN
    ta.configure_gc(freeze=False)
    assert calls[-1] == "unfreeze" and not dt._GC_FROZEN
  finally:
    dt._GC_POLICY.update(saved[0])
    dt._GC_FROZEN, dt._gc_cost_seconds = saved[1], saved[2]


def test_mera_layer_lowering_keeps_the_big_intermediate_in_place():
  """Host dry run of the binary-MERA layer at chi = 32 (tools/mera_trace.py: the real lowering code on shape-only
  tensors): every contraction of both placements is ONE in-place view GEMM, the 68 GB intermediate is never permuted
  -- the planner's layout hints win over an available view when the operand is small against the result -- and the
  K1 passes that remain move 4.3 GB tensors only."""
  import re
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = subprocess.run([sys.executable, os.path.join(root, "tools", "mera_trace.py"), "--chi", "32"], capture_output=True,
                       text=True, timeout=300, check=True).stdout
  assert "trace stopped" not in out, out
  gemms = re.findall(r"view GEMM M=(\d+) N=(\d+) K=(\d+)", out)
  assert len(gemms) == 12, out                                   # six contractions per placement, none on the fallback
  moved = [float(x) for x in re.findall(r"permute .* ([0-9.]+) GB moved", out)]
  assert moved and max(moved) < 10.0 and sum(moved) < 30.0, moved
  # the two large products of a placement: 2^20 x 2^15 x 2^10 and 2^10 x 2^15 x 2^20
  big = sorted((int(m), int(n), int(k)) for m, n, k in gemms if int(m) * int(n) * int(k) >= 2**45)
  assert big == [(1024, 32768, 1048576)] * 2 + [(1048576, 32768, 1024)] * 2, big
