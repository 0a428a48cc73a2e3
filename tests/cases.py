"""Backend-agnostic drivers for the golden cases.

Each ``run_*`` function executes one fixture case through the host layer of
``tensornetwork_amd`` (ncon / network / contractors) on the backend object it is
given -- the NumPy oracle backend in the CPU suite, ``HipBackend`` in the GPU
suite -- and returns host arrays for comparison with the reference's output.
"""
import numpy as np

import tensornetwork_amd as ta
from tensornetwork_amd import contractors, network


def to_host(x):
  return np.asarray(x)


def tol(dtype, scale=1.0):
  """Comparison tolerance vs the reference's NumPy result:
  fp32 1e-5-class, fp64/complex128 1e-12-class, relative to the result's scale."""
  dt = np.dtype(dtype)
  if dt in (np.dtype(np.float32), np.dtype(np.complex64)):
    return dict(rtol=2e-5, atol=2e-5 * scale)
  if dt == np.dtype(np.float16):
    return dict(rtol=5e-2, atol=5e-2 * scale)
  return dict(rtol=1e-11, atol=1e-11 * scale)


def assert_close(actual, expected, scale=None):
  actual, expected = to_host(actual), np.asarray(expected)
  assert actual.shape == expected.shape, (actual.shape, expected.shape)
  if scale is None:
    scale = float(np.max(np.abs(expected))) if expected.size else 1.0
  np.testing.assert_allclose(actual, expected, **tol(expected.dtype, max(scale, 1e-30)))


# tolerance of a bf16 / f16 GEMM result against float64 on the ROUNDED inputs (bf16-out rtol 2^-8): the
# result is rounded once to the storage type (2^-9 / 2^-12 relative, tested at twice that) and the fp32 accumulation
# over K terms sits far below 2^-10 / 2^-13 of the rms entry.  No sqrt(K) factor: a kernel that loses bits fails.
HALF_TOL = {"bf16": (2.0**-8, 2.0**-10), "f16": (2.0**-11, 2.0**-13)}


def assert_half_gemm_close(out, ref, kind, err_msg=""):
  """|out - ref| <= rel |ref| + abs_rms rms(ref) elementwise (the rule of bench.verify_pair), kind in HALF_TOL."""
  rel, rms_part = HALF_TOL[kind]
  out, ref = np.asarray(out).astype(np.float64), np.asarray(ref).astype(np.float64)
  assert out.shape == ref.shape, (out.shape, ref.shape, err_msg)
  rms = float(np.sqrt(np.mean(ref**2))) if ref.size else 0.0
  tolv = rel * np.abs(ref) + rms_part * rms
  err = np.abs(out - ref)
  worst = float((err / np.maximum(tolv, 1e-300)).max()) if ref.size else 0.0
  assert worst <= 1.0, f"{err_msg}: max err / tol = {worst:.3f} (tol = {rel:g} |ref| + {rms_part:g} rms, rms = {rms:g})"


def host64(t):
  return np.asarray(t).astype(np.float64)


def verify_pair(be, A, B, out, layout, n_side=32, seed=0):
  """>= 1024 sampled entries of a rank-4 x rank-4 contraction against float64 dot products of the DEVICE
  operands.  Only the needed slabs are read back: n_side (row pair) slabs of A, n_side (column
  pair) slabs of B and n_side slabs of the result.
    L0: C[i0,i1,j2,j3] = sum_{k2,k3} A[i0,i1,k2,k3] B[k2,k3,j2,j3]
    L1: C[i0,i2,j1,j3] = sum_{k1,k3} A[i0,k1,i2,k3] B[k3,j1,k1,j3]
  Tolerance: bf16 rounding of the result (2^-9 relative, tested at 2^-8) + 2^-10 of the rms entry (fp32
  accumulation over K terms is far below that)."""
  rng = np.random.default_rng(seed)
  sa, sb = A.shape, B.shape
  if layout == "L0":
    rdims, cdims = (sa[0], sa[1]), (sb[2], sb[3])
  else:
    rdims, cdims = (sa[0], sa[2]), (sb[1], sb[3])
  rows = [(int(rng.integers(rdims[0])), int(rng.integers(rdims[1]))) for _ in range(n_side)]
  cols = [(int(rng.integers(cdims[0])), int(rng.integers(cdims[1]))) for _ in range(n_side)]
  # always include the four corners of the output (first / last tile of the launch)
  rows[0], rows[-1] = (0, 0), (rdims[0] - 1, rdims[1] - 1)
  cols[0], cols[-1] = (0, 0), (cdims[0] - 1, cdims[1] - 1)
  sl = slice(None)
  if layout == "L0":
    a_rows = np.stack([host64(be.getitem(A, (r0, r1))).reshape(-1) for r0, r1 in rows])
    b_cols = np.stack([host64(be.getitem(B, (sl, sl, c0, c1))).reshape(-1) for c0, c1 in cols])
  else:
    a_rows = np.stack([host64(be.getitem(A, (r0, sl, r1, sl))).reshape(-1) for r0, r1 in rows])
    b_cols = np.stack([host64(be.getitem(B, (sl, c0, sl, c1))).T.reshape(-1) for c0, c1 in cols])
  ref = a_rows @ b_cols.T                                            # (n_side, n_side) float64
  got = np.empty_like(ref)
  for i, (r0, r1) in enumerate(rows):
    slab = host64(be.getitem(out, (r0, r1)))                         # [c0, c1]
    got[i] = [slab[c0, c1] for c0, c1 in cols]
  rms = float(np.sqrt(np.mean(ref**2)))
  tol = 2.0**-8 * np.abs(ref) + 2.0**-10 * rms
  err = np.abs(got - ref)
  return {"entries": int(ref.size), "max_abs_err": float(err.max()), "rms_ref": rms,
          "max_err_over_tol": float((err / tol).max()), "tol": "2^-8 |ref| + 2^-10 rms(ref)",
          "ok": bool((err <= tol).all())}


def run_tensordot(be, g, case):
  a, b = be.convert_to_tensor(g[case["a"]]), be.convert_to_tensor(g[case["b"]])
  return be.tensordot(a, b, case["axes"])


def run_transpose(be, g, case):
  return be.transpose(be.convert_to_tensor(g[case["x"]]), case["perm"])


def run_ncon(be, g, case):
  tensors = [g[name] for name in case["tensors"]]
  return ta.ncon(tensors, case["structure"], con_order=case["con_order"], out_order=case["out_order"],
                 backend=be)


def run_contract_between(be, g, case):
  kind = case["kind"]
  if kind == "between":
    a, b = network.Node(g[case["a"]], backend=be), network.Node(g[case["b"]], backend=be)
    for x, y in case["connect"]:
      network.connect(a[x], b[y])
    order = None
    if case["order"] is not None:
      order = [(a if who == "a" else b)[ax] for who, ax in case["order"]]
    return network.contract_between(a, b, output_edge_order=order).tensor
  if kind == "physics":
    a, b, c = (network.Node(g[case[k]], backend=be) for k in "abc")
    e1 = network.connect(a[2], b[0])
    e2 = network.connect(c[0], a[3])
    e3 = network.connect(b[1], c[1])
    network.contract(e1)
    network.contract(e2)
    return network.contract(e3).tensor
  if kind == "trace":
    t = network.Node(g[case["a"]], backend=be)
    network.connect(t[0], t[2])
    return network.contract_between(t, t).tensor
  if kind == "outer":
    return network.outer_product(network.Node(g[case["a"]], backend=be),
                                 network.Node(g[case["b"]], backend=be)).tensor
  raise ValueError(kind)


def run_split(be, g, case):
  """Returns (left_node, right_node, trun_vals, reconstruction)."""
  node = network.Node(g[case["x"]], backend=be)
  kw = dict(case["kw"])
  left, right, trun = network.split_node(node, [node[i] for i in case["left"]],
                                         [node[i] for i in case["right"]], **kw)
  recon = network.contract_between(left, right)
  return left, right, trun, recon.tensor


def run_contractor(be, g, case):
  kind = case["kind"]
  if kind == "mps_overlap":
    kets = [g[name] for name in case["kets"]]
    nk = [network.Node(k, backend=be) for k in kets]
    nb = [network.Node(np.conj(k), backend=be) for k in kets]
    n = len(kets)
    for i in range(n):
      network.connect(nk[i][1], nb[i][1])
      if i + 1 < n:
        network.connect(nk[i][2], nk[i + 1][0])
        network.connect(nb[i][2], nb[i + 1][0])
    network.connect(nk[0][0], nb[0][0])
    network.connect(nk[-1][2], nb[-1][2])
    return contractors.greedy(nk + nb).tensor
  if kind == "regular":
    nodes = [network.Node(g[name], backend=be) for name in case["tensors"]]
    for x, sx, y, sy in case["edges"]:
      network.connect(nodes[x][sx], nodes[y][sy])
    return contractors.greedy(nodes).tensor
  if kind == "open3":
    a, b, c = (network.Node(g[case[k]], backend=be) for k in "abc")
    network.connect(a[2], b[0])
    network.connect(b[2], c[0])
    network.connect(a[1], c[1])
    return getattr(contractors, case["alg"])([a, b, c], output_edge_order=[c[2], a[0], b[1]]).tensor
  raise ValueError(kind)


def run_misc(be, g, case):
  """dict name -> result for the helper methods ncon / split_node rely on."""
  x, v, w, m = (be.convert_to_tensor(g[case[k]]) for k in "xvwm")
  return {
      "brm": be.broadcast_right_multiplication(x, v),
      "blm": be.broadcast_left_multiplication(w, x),
      "sum12": be.sum(x, axis=(1, 2)),
      "sum0": be.sum(x, axis=(0,)),
      "sum02": be.sum(x, axis=(0, 2)),
      "trace": be.trace(m),
      "trace1": be.trace(m, offset=1),
      "matmul": be.matmul(m, m),
      "outer": be.outer_product(v, w),
      "diagflat": be.diagflat(v),
      "conj": be.conj(x),
      "add": be.addition(x, x),
      "mul": be.multiply(x, v),
      "slice": be.slice(x, (1, 0, 2), (2, 3, 2)),
  }


# ----------------------------------------------------------- QR / dense linalg
def check_qr_case(be, g, case, tight=True):
  """qr and rq of one fixture through backend `be`.  R is compared with the reference
  element-wise (Householder QR is unique up to the reflector sign convention, which the
  HIP kernel shares with LAPACK); Q through orthonormality and reconstruction, plus
  element-wise where the matrix has full column rank."""
  x = g[case["x"]]
  scale = float(np.max(np.abs(x))) + 1e-30
  q, r = be.qr(be.convert_to_tensor(x), case["pivot"], case["nnd"])
  q, r = to_host(q), to_host(r)
  assert q.shape == g[case["q"]].shape and r.shape == g[case["r"]].shape
  k = q.shape[-1]
  qm, rm = q.reshape(-1, k), r.reshape(k, -1)
  t = tol(x.dtype, scale)
  np.testing.assert_allclose(qm @ rm, x.reshape(qm.shape[0], rm.shape[1]), rtol=0, atol=10 * t["atol"])
  np.testing.assert_allclose(qm.conj().T @ qm, np.eye(k), rtol=0, atol=10 * tol(x.dtype)["atol"])
  assert np.allclose(rm, np.triu(rm))
  if tight:
    assert_close(r, g[case["r"]], scale=scale * np.sqrt(qm.shape[0]))
    assert_close(q, g[case["q"]], scale=10.0)
  rr, qq = be.rq(be.convert_to_tensor(x), case["pivot"], case["nnd"])
  rr, qq = to_host(rr), to_host(qq)
  assert rr.shape == g[case["rq_r"]].shape and qq.shape == g[case["rq_q"]].shape
  k = rr.shape[-1]
  rm, qm = rr.reshape(-1, k), qq.reshape(k, -1)
  np.testing.assert_allclose(rm @ qm, x.reshape(rm.shape[0], qm.shape[1]), rtol=0, atol=10 * t["atol"])
  np.testing.assert_allclose(qm @ qm.conj().T, np.eye(k), rtol=0, atol=10 * tol(x.dtype)["atol"])
  if tight:
    assert_close(rr, g[case["rq_r"]], scale=scale * np.sqrt(qm.shape[1]))
    assert_close(qq, g[case["rq_q"]], scale=10.0)


def check_split_qr_case(be, g, case):
  x = g[case["x"]]
  a = network.Node(be.convert_to_tensor(x), backend=be)
  q, r = network.split_node_qr(a, [a[i] for i in case["left"]], [a[i] for i in case["right"]])
  assert_close(q.tensor, g[case["q"]], scale=10.0)
  assert_close(r.tensor, g[case["r"]], scale=float(np.max(np.abs(x))) * 10)
  assert q.edges[-1] is r.edges[0]
  b = network.Node(be.convert_to_tensor(x), backend=be)
  r2, q2 = network.split_node_rq(b, [b[i] for i in case["left"]], [b[i] for i in case["right"]])
  assert_close(r2.tensor, g[case["rq_r"]], scale=float(np.max(np.abs(x))) * 10)
  assert_close(q2.tensor, g[case["rq_q"]], scale=10.0)
  # the split network contracts back to the original tensor
  back = network.contract_between(r2, q2)
  assert_close(back.tensor, np.transpose(x, case["left"] + case["right"]), scale=float(np.max(np.abs(x))) * 10)


def check_linalg_case(be, g, case):
  if "h" in case:
    h = g[case["h"]]
    n = h.shape[0]
    w, v = be.eigh(be.convert_to_tensor(h))
    w, v = to_host(w), to_host(v)
    scale = float(np.max(np.abs(h))) * max(n, 1) ** 0.5 + 1e-30
    assert_close(w, g[case["w"]], scale=scale)
    t = tol(h.dtype, scale)
    np.testing.assert_allclose(h @ v, v * w, rtol=0, atol=30 * t["atol"])
    np.testing.assert_allclose(v.conj().T @ v, np.eye(n), rtol=0, atol=30 * tol(h.dtype)["atol"])
    assert not np.iscomplexobj(w)
  if "inv" in case:
    a = g[case["a"]]
    got = to_host(be.inv(be.convert_to_tensor(a)))
    assert_close(got, g[case["inv"]], scale=float(np.max(np.abs(g[case["inv"]]))) * 10)
  if "expm" in case:
    e = g[case["e"]]
    got = to_host(be.expm(be.convert_to_tensor(e)))
    ref = g[case["expm"]]
    assert_close(got, ref, scale=float(np.max(np.abs(ref))) * 100)


# ------------------------------------------------------------------ MPS / DMRG
def xxz_dense(n, jz, jxy, bz):
  """Dense XXZ Hamiltonian built independently of the MPO (Kronecker products)."""
  sz = np.diag([-0.5, 0.5])
  sp = np.array([[0.0, 0.0], [1.0, 0.0]])
  sm = sp.T
  def op(o, i):
    mats = [np.eye(2)] * n
    mats[i] = o
    out = mats[0]
    for m in mats[1:]:
      out = np.kron(out, m)
    return out
  h = np.zeros((2**n, 2**n))
  for i in range(n - 1):
    h += jz * op(sz, i) @ op(sz, i + 1) + jxy / 2 * (op(sp, i) @ op(sm, i + 1) + op(sm, i) @ op(sp, i + 1))
  for i in range(n):
    h += bz * op(sz, i)
  return h


def check_svd_case(be, g, case):
  """svd of one fixture (complex-aware): spectrum and discarded values element-wise, factors through
  orthonormality and the truncated reconstruction (vectors are unique only up to phases)."""
  x = g[case["x"]]
  kw = case["kw"]
  u, s, vh, rest = be.svd(be.convert_to_tensor(x), case["pivot"], kw.get("max_singular_values"),
                          kw.get("max_truncation_error"), kw.get("relative", False))
  u, s, vh, rest = to_host(u), to_host(s), to_host(vh), to_host(rest)
  ur, sr, vr, rr = g[case["u"]], g[case["s"]], g[case["vh"]], g[case["rest"]]
  assert u.shape == ur.shape and vh.shape == vr.shape and s.shape == sr.shape and rest.shape == rr.shape
  assert s.dtype == sr.dtype and u.dtype == ur.dtype
  scale = float(np.max(np.abs(x))) * np.sqrt(max(x.shape)) + 1e-30
  assert_close(s, sr, scale=scale)
  assert_close(rest, rr, scale=scale)
  k = s.shape[0]
  um, vm = u.reshape(-1, k), vh.reshape(k, -1)
  t = tol(x.dtype)
  np.testing.assert_allclose(um.conj().T @ um, np.eye(k), rtol=0, atol=20 * t["atol"])
  np.testing.assert_allclose(vm @ vm.conj().T, np.eye(k), rtol=0, atol=20 * t["atol"])
  ref = (ur.reshape(-1, k) * sr) @ vr.reshape(k, -1)
  np.testing.assert_allclose((um * s) @ vm, ref, rtol=0, atol=50 * tol(x.dtype, scale)["atol"])


# ------------------------------------------------------------------ MPS measurement goldens
MPS_GOLDEN_TAGS = ("f64c3", "f64c0", "c128c7")


def load_mps_golden():
  import os
  return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_mps.npz"))


def check_mps_golden_case(be, g, tag, atol):
  """Everything tests/golden/make_golden_mps.py recorded from the reference's FiniteMPS, recomputed by
  tensornetwork_amd.mps.FiniteMPS on backend `be` from the same (already canonical) tensors."""
  from tensornetwork_amd import mps as tmps
  n, center = [int(x) for x in g[f"{tag}_meta"]]
  state = tmps.FiniteMPS([be.convert_to_tensor(g[f"{tag}_t{k}"]) for k in range(n)], be,
                         center_position=center, canonicalize=False)
  host = lambda x: np.asarray(x)
  assert state.center_position == center
  assert state.physical_dimensions == [g[f"{tag}_t{k}"].shape[1] for k in range(n)]
  assert [state.bond_dimension(k) for k in range(n + 1)] == state.bond_dimensions
  sites = list(range(n))
  ls = state.left_envs(sites + [n])
  rs = state.right_envs([-1] + sites)
  assert sorted(ls) == list(range(n + 1)) and sorted(rs) == list(range(-1, n))
  for k, v in ls.items():
    np.testing.assert_allclose(host(v), g[f"{tag}_L{k}"], atol=atol)
  for k, v in rs.items():
    np.testing.assert_allclose(host(v), g[f"{tag}_R{k}"], atol=atol)
  assert state.left_envs([]) == {} and state.right_envs([]) == {}
  np.testing.assert_allclose(host(state.apply_transfer_operator(3, "left", be.convert_to_tensor(g[f"{tag}_ml"]))),
                             g[f"{tag}_tl"], atol=atol)
  np.testing.assert_allclose(host(state.apply_transfer_operator(3, -1, be.convert_to_tensor(g[f"{tag}_mr"]))),
                             g[f"{tag}_tr"], atol=atol)
  op1, op2 = be.convert_to_tensor(g[f"{tag}_op1"]), be.convert_to_tensor(g[f"{tag}_op2"])
  np.testing.assert_allclose(state.measure_local_operator([op1] * n, sites), g[f"{tag}_local"], atol=atol)
  assert state.center_position == center           # measuring does not move the gauge
  for s1 in (0, 3, n - 1):
    np.testing.assert_allclose(state.measure_two_body_correlator(op1, op2, s1, sites), g[f"{tag}_corr{s1}"],
                               atol=atol)
  np.testing.assert_allclose(state.measure_two_body_correlator(op1, op2, 4, [6, 1, 4, 6]), g[f"{tag}_corr_sub"],
                             atol=atol)
  np.testing.assert_allclose(state.check_canonical(), float(g[f"{tag}_canon"]), atol=atol)
  state.apply_one_site_gate(op2, 2)
  np.testing.assert_allclose(host(state.get_tensor(2)), g[f"{tag}_gated2"], atol=atol)
  np.testing.assert_allclose(state.check_canonical(), float(g[f"{tag}_canon_after"]), atol=atol * 10)


# ------------------------------------------------------------------ graph surgery (flatten / split / ...)
def check_graph_surgery(be, rtol):
  """The reference's own graph-operation tests, restated for a backend object `be`:
  network_test.py:278-330 (flatten), 359-428 (split_edge), 431-470 (parallel / flatten_between / all),
  587-594 + tensornetwork_test.py:661-720 (remove_node), 421-500 (flatten keeps results),
  network_operations_test.py:331-374 (reduced_density), 471-560 (neighbors, redirect),
  network_components_free_test.py:1120-1150 (disconnect)."""
  import pytest
  import tensornetwork_amd as tn
  rng = np.random.default_rng(77)
  N = lambda arr, **kw: tn.Node(np.asarray(arr, dtype=np.float64), backend=be, **kw)
  host = lambda node: np.asarray(node.tensor)

  # flatten: trace edges
  a, c = N(np.zeros((2, 3, 4, 3, 5, 5))), N(np.zeros((2, 4)))
  e1, e2 = tn.connect(a[1], a[3]), tn.connect(a[4], a[5])
  x1, x2 = tn.connect(a[0], c[0]), tn.connect(c[1], a[2])
  new = tn.flatten_edges([e1, e2], "New Edge")
  tn.check_correct({a, c})
  assert a.shape == (2, 4, 15, 15) and a.edges == [x1, x2, new, new] and new.name == "New Edge"
  # flatten: standard
  a, b = N(np.zeros((2, 3, 5)), name="A"), N(np.zeros((2, 3, 4, 5)), name="B")
  e1, e2 = tn.connect(a[0], b[0], "Edge_1_1"), tn.connect(a[2], b[3], "Edge_2_3")
  ea1, eb1, eb2 = a[1], b[1], b[2]
  new = tn.flatten_edges([e1, e2], new_edge_name="New Edge")
  assert a.shape == (3, 10) and b.shape == (3, 4, 10)
  assert a.edges == [ea1, new] and b.edges == [eb1, eb2, new]
  tn.check_correct({a, b})
  # flatten: dangling
  a = N(np.zeros((2, 3, 4, 5)), name="A")
  d1, d2, d3, d4 = a[0], a[1], a[2], a[3]
  new = tn.flatten_edges([d1, d3], new_edge_name="New Edge")
  assert a.shape == (3, 5, 8) and a.edges == [d2, d4, new] and new.name == "New Edge"
  tn.check_correct({a})
  assert tn.flatten_edges([d2]) is d2
  with pytest.raises(ValueError, match="At least 1 edge"):
    tn.flatten_edges([])
  a, b, c = N(np.eye(2)), N(np.eye(2)), N(np.eye(2))
  e1, e2 = tn.connect(a[0], b[0]), tn.connect(a[1], c[0])
  with pytest.raises(ValueError, match="do not share the same nodes"):
    tn.flatten_edges([e1, e2])

  # flatten keeps contraction results (tensornetwork_test.py:421-500), values through the backend
  ta_, tb_ = rng.standard_normal((3, 5, 4, 6)), rng.standard_normal((6, 5, 7, 3))
  a, b = N(ta_), N(tb_)
  tn.connect(a[0], b[3]); tn.connect(b[1], a[1]); tn.connect(a[3], b[0])
  want = np.einsum("abcd,dbea->ce", ta_, tb_)
  flat = tn.flatten_edges_between(a, b)
  tn.check_correct({a, b})
  assert flat.dimension == 90 and a.shape == (4, 90) and b.shape == (7, 90)
  np.testing.assert_allclose(host(tn.contract(flat)), want, rtol=rtol, atol=rtol)
  tt = rng.standard_normal((3, 4, 3, 5, 4))
  a = N(tt)
  t1, t2 = tn.connect(a[0], a[2]), tn.connect(a[4], a[1])
  flat = tn.flatten_edges([t1, t2])
  assert a.shape == (5, 12, 12)
  np.testing.assert_allclose(host(tn.contract(flat)), np.einsum("abaeb->e", tt), rtol=rtol, atol=rtol)

  # split_edge: standard, then contract_between agrees with the unsplit network
  ta_, tb_ = rng.standard_normal((6, 3, 5)), rng.standard_normal((2, 4, 6, 3))
  a, b = N(ta_, name="A"), N(tb_, name="B")
  e1, e2 = tn.connect(a[0], b[2], "Edge_1_1"), tn.connect(a[1], b[3], "Edge_1_2")
  ea2, eb0, eb1 = a[2], b[0], b[1]
  shape, names = (2, 1, 3), ["New Edge 2", "New Edge 1", "New Edge 3"]
  new_edges = tn.split_edge(e1, shape, names)
  assert a.shape == (3, 5) + shape and b.shape == (2, 4, 3) + shape
  assert a.edges == [e2, ea2, *new_edges] and b.edges == [eb0, eb1, e2, *new_edges]
  assert [e.dimension for e in new_edges] == list(shape) and [e.name for e in new_edges] == names
  tn.check_correct({a, b})
  np.testing.assert_allclose(host(tn.contract_between(a, b)), np.einsum("abc,deab->cde", ta_, tb_),
                             rtol=rtol, atol=rtol)
  # split_edge: dangling, trivial, trace, mismatch
  a = N(np.zeros((2, 10, 4, 5)), name="A")
  d1, d2, d3, d4 = a[0], a[1], a[2], a[3]
  new_edges = tn.split_edge(d2, (2, 5), ["New Edge 2", "New Edge 5"])
  assert a.shape == (2, 4, 5, 2, 5) and a.edges == [d1, d3, d4, *new_edges]
  assert [e.dimension for e in new_edges] == [2, 5] and [e.name for e in new_edges] == ["New Edge 2", "New Edge 5"]
  tn.check_correct({a})
  assert tn.split_edge(d1, (2,)) == [d1]
  tt = rng.standard_normal((6, 3, 6))
  a = N(tt)
  tr = tn.connect(a[0], a[2])
  parts = tn.split_edge(tr, (2, 3))
  assert a.shape == (3, 2, 3, 2, 3) and all(p.is_trace() for p in parts)
  tn.check_correct({a})
  np.testing.assert_allclose(host(tn.contract_trace_edges(a)), np.einsum("aba->b", tt), rtol=rtol, atol=rtol)
  a = N(np.eye(5))
  with pytest.raises(ValueError, match="cannot be split according to shape"):
    tn.split_edge(tn.connect(a[0], a[1]), (2, 2))

  # parallel edges, flatten_edges_between, flatten_all_edges
  a, b = N(np.ones((2,) * 5)), N(np.ones((2,) * 5))
  es = {tn.connect(a[i], b[i]) for i in (0, 1, 3)}
  assert all(tn.get_parallel_edges(e) == es for e in es)
  a, b = N(np.ones((3, 4, 5))), N(np.ones((5, 4, 3)))
  tn.connect(a[0], b[2]); tn.connect(a[1], b[1]); tn.connect(a[2], b[0])
  tn.flatten_edges_between(a, b)
  tn.check_correct({a, b})
  np.testing.assert_array_equal(host(a), np.ones(60)); np.testing.assert_array_equal(host(b), np.ones(60))
  assert tn.flatten_edges_between(N(np.ones(3)), N(np.ones(3))) is None
  a, b, c = N(np.ones((3, 3, 5, 6, 2, 2))), N(np.ones((5, 6, 7))), N(np.ones((7,)))
  tn.connect(a[0], a[1]); tn.connect(a[4], a[5]); tn.connect(a[2], b[0]); tn.connect(a[3], b[1])
  ok = tn.connect(b[2], c[0])
  flat = tn.flatten_all_edges([a, b, c])
  tn.check_correct({a, b, c})
  assert len(flat) == 3 and ok in flat and a.shape == (6, 6, 30) and b.shape == (7, 30)
  np.testing.assert_allclose(host(tn.contractors.greedy([a, b, c])), 6 * 30 * 7, rtol=rtol)

  # disconnect / remove_node / neighbors / redirect / checks
  a, b = N(np.eye(2)), N(np.eye(2))
  e = tn.connect(a[0], b[0], name="bond")
  l, r = tn.disconnect(e)
  assert l.is_dangling() and r.is_dangling() and a[0] is l and b[0] is r
  assert l.name == "__disconnected_edge1_of_bond__" and r.name == "__disconnected_edge2_of_bond__"
  with pytest.raises(ValueError, match="Cannot break dangling edge"):
    tn.disconnect(a[1])
  a, b = N(np.eye(2)), N(np.eye(2))
  tn.connect(a[0], b[0])
  by_name, by_axis = tn.remove_node(b)
  assert by_name == {"0": a[0]} and by_axis == {0: a[0]} and a[0].is_dangling()
  a, b, c = N(np.ones((2, 2, 2)), axis_names=["x", "y", "z"]), N(np.ones((2, 2))), N(np.ones((2, 2, 2)))
  tn.connect(a["x"], b[0]); tn.connect(a["y"], c[1]); tn.connect(c[0], c[2])
  by_name, by_axis = tn.remove_node(a)
  assert set(by_name) == {"x", "y"} and by_name["x"] is b[0] and by_axis[1] is c[1] and c[0].is_trace()
  a, b, c = N(np.ones((2, 2, 2))), N(np.ones((2, 2, 2))), N(np.ones((2, 2, 2)))
  tn.connect(a[0], b[0]); tn.connect(a[1], b[1]); tn.connect(a[2], c[0]); tn.connect(c[1], c[2])
  assert tn.get_neighbors(a) == [b, c] and tn.get_neighbors(c) == [a]
  assert tn.get_all_nodes(tn.get_all_edges([a])) == {a, b, c}
  assert len(tn.get_all_nondangling([a, b, c])) == 4 and tn.get_all_dangling([a, b, c]) == [b[2]]
  tn.check_connected([a, b, c])
  with pytest.raises(ValueError, match="Non-connected graph"):
    tn.check_connected([a, N(np.eye(2))])
  a, b, c = N(np.ones((2, 3))), N(np.ones((3, 4))), N(np.ones((2, 3)))
  e = tn.connect(a[1], b[0])
  tn.redirect_edge(e, c, a)
  tn.check_correct({b, c}); tn.check_correct({a}, check_connections=False)
  assert c[1] is e and a[1].is_dangling() and set(map(id, e.get_nodes())) == {id(b), id(c)}
  with pytest.raises(ValueError, match="is not pointing to old_node"):
    tn.redirect_edge(e, a, a)
  a, b = N(np.eye(2)), N(np.eye(2))
  e = tn.connect(a[0], b[0])
  e.axis1 = 1                                       # corrupt on purpose (network_operations_test.py:424-437)
  with pytest.raises(ValueError, match="does not point to"):
    tn.check_correct({a, b})

  # reduced_density + replicate_nodes + from_topology
  ta_, tb_ = rng.standard_normal((2, 3, 4)), rng.standard_normal((4, 5))
  a, b = N(ta_, name="A"), N(tb_, name="B")
  tn.connect(a[2], b[0])
  keep0, keep1, traced = a[0], a[1], b[1]
  node_map, edge_map = tn.reduced_density([traced])
  assert set(node_map) == {a, b} and not edge_map[traced].is_dangling()
  with pytest.raises(ValueError, match="must only include dangling edges"):
    tn.reduced_density([edge_map[traced]])
  nodes = [a, b, node_map[a], node_map[b]]
  tn.check_correct(nodes)
  rho = tn.contractors.greedy(nodes, output_edge_order=[keep0, keep1, edge_map[keep0], edge_map[keep1]])
  psi = np.einsum("abc,cd->abd", ta_, tb_)
  np.testing.assert_allclose(host(rho), np.einsum("abd,efd->abef", psi, psi), rtol=rtol, atol=rtol)
  a, b = N(ta_), N(tb_)
  tn.connect(a[2], b[0])
  ra, rb = tn.replicate_nodes([a, b])
  assert ra is not a and tn.get_shared_edges(ra, rb) and not tn.get_shared_edges(ra, b)
  x, y, z = tn.from_topology("abc,bceg,adef", [np.ones((2,) * n) for n in (3, 4, 4)], backend=be)
  assert x.axis_names == ["a", "b", "c"] and y.axis_names == ["b", "c", "e", "g"]
  assert x["a"] is z["a"] and x["b"] is y["b"] and x["c"] is y["c"] and y["e"] is z["e"]
  assert z["d"].is_dangling() and z["f"].is_dangling() and y["g"].is_dangling()
  with pytest.raises(ValueError, match="mismatched"):
    tn.from_topology("ab,bc", [np.ones((2, 2))], backend=be)
  with pytest.raises(ValueError, match="does not match shape"):
    tn.from_topology("abc", [np.ones((2, 2))], backend=be)

  # CopyNode / contract_copy_node (tensornetwork_test.py:592-660) and Node arithmetic (network_components.py:586-631)
  a, b, c, d = N([1, 2, 3]), N([10, 20, 30]), N([5, 6, 7]), N([1, -1, 1])
  cn = tn.CopyNode(rank=4, dimension=3, backend=be)
  assert cn.shape == (3, 3, 3, 3) and cn.dtype == np.float64 and cn._dense is None
  es = [tn.connect(x[0], cn[i]) for i, x in enumerate((a, b, c, d))]
  assert set(cn.get_partners()) == {a, b, c, d}
  np.testing.assert_allclose(np.asarray(cn.compute_contracted_tensor()), 50 - 240 + 630, rtol=rtol)
  assert cn._dense is None                                   # never materialised by the einsum route
  for e in es:
    val = tn.contract(e)                                     # the dense route: one edge at a time
  np.testing.assert_allclose(host(val), 50 - 240 + 630, rtol=rtol)
  np.testing.assert_array_equal(tn.CopyNode.make_copy_tensor(3, 2, np.float32), np.einsum("ij,jk->ijk", np.eye(2), np.eye(2)))
  a, b, cn = N(np.diag([1.0, 2, 3])), N([10, 20, 30]), tn.CopyNode(rank=3, dimension=3, backend=be)
  tn.connect(a[0], cn[0]); tn.connect(a[1], cn[1]); tn.connect(b[0], cn[2])
  np.testing.assert_allclose(np.asarray(cn.compute_contracted_tensor()), 10 + 40 + 90, rtol=rtol)
  a, b, c = N([[1, 2, 3], [10, 20, 30]]), N([[2, 1, 1], [2, 2, 2]]), N([3, 4, 4])
  cn = tn.CopyNode(rank=3, dimension=3, backend=be)
  tn.connect(a[0], b[0]); tn.connect(a[1], cn[0]); tn.connect(b[1], cn[1]); tn.connect(c[0], cn[2])
  n = tn.contract_copy_node(cn)
  assert len(n.edges) == 2 and n.edges[0] is n.edges[1] and all(e.is_dangling() for e in cn.edges)
  np.testing.assert_allclose(host(tn.contract_parallel(n.edges[0])), 26 + 460, rtol=rtol)
  with pytest.raises(ValueError, match="dangling edges"):
    tn.CopyNode(rank=2, dimension=2, backend=be).compute_contracted_tensor()
  x = N(rng.standard_normal((2, 3)), name="x")
  xh = host(x)
  y = (x * 3 + x - 1) / 2
  np.testing.assert_allclose(host(y), (xh * 3 + xh - 1) / 2, rtol=rtol)
  np.testing.assert_allclose(host(x * x - x / (x * x + 1)), xh * xh - xh / (xh * xh + 1), rtol=rtol)
  assert y.name == "x" and all(e.is_dangling() for e in y.edges) and y is not x
  with pytest.raises(TypeError, match="Operand should be one of"):
    x + "1"  # pylint: disable=pointless-statement
  with pytest.raises(NotImplementedError):
    cn + 1  # pylint: disable=pointless-statement

  # bucket elimination over copy tensors (bucket_contractor_test.py:24-98): CNOT = COPY on the control + XOR on the target
  def add_cnot(q0, q1):
    control = tn.CopyNode(rank=3, dimension=2, backend=be)
    target = N([[[1, 0], [0, 1]], [[0, 1], [1, 0]]])
    tn.connect(q0, control[0]); tn.connect(q1, target[0]); tn.connect(control[1], target[1])
    return control, control[2], target[2]
  q0_in, q1_in, q0_out, q1_out = N([0, 1]), N([0, 1]), N([0, 1]), N([1, 0])            # |11> -> |10>
  cnot, t0, t1 = add_cnot(q0_in[0], q1_in[0])
  tn.connect(t0, q0_out[0]); tn.connect(t1, q1_out[0])
  net = tn.contractors.bucket([q0_in, q1_in, q0_out, q1_out, cnot], (cnot,))
  np.testing.assert_allclose(host(tn.contractors.greedy(net)), 1.0, rtol=rtol)
  q0_in, q1_in, q0_out, q1_out = N([0.6, 0.8]), N([1, 0]), N([1, 0]), N([0.6, 0.8])    # three CNOTs = SWAP
  c1, a0, a1 = add_cnot(q0_in[0], q1_in[0])
  c2, b1, b0 = add_cnot(a1, a0)
  c3, d0, d1 = add_cnot(b0, b1)
  tn.connect(d0, q0_out[0]); tn.connect(d1, q1_out[0])
  net = tn.contractors.bucket([q0_in, q0_out, q1_in, q1_out, c1, c2, c3], (c1, c2, c3))
  np.testing.assert_allclose(host(tn.contractors.greedy(net)), 1.0, rtol=rtol)


INFINITE_MPS_GOLDEN_TAGS = ("inf_f64", "inf_c128")


def check_infinite_mps_golden_case(be, g, tag, rtol):
  """InfiniteMPS on backend `be` vs the reference's InfiniteMPS on the same unit cell
  (tests/golden/make_golden_mps.py:gen_infinite; infinite_mps_test.py:57-91): unit-cell transfer
  operator, its dominant eigenpair, Schmidt-canonical form."""
  from tensornetwork_amd import mps as tmps
  n = int(g[f"{tag}_meta"][0])
  imps = tmps.InfiniteMPS([be.convert_to_tensor(g[f"{tag}_t{k}"]) for k in range(n)], be, center_position=0)
  dtype = g[f"{tag}_t0"].dtype
  m = be.convert_to_tensor(g[f"{tag}_m"])
  for direction, key in (("left", "uc_l"), (-1, "uc_r")):
    ref = g[f"{tag}_{key}"]
    np.testing.assert_allclose(np.asarray(imps.unit_cell_transfer_operator(direction, m)), ref,
                               rtol=rtol, atol=rtol * np.abs(ref).max())
  np.random.seed(1)
  eta, l = imps.transfer_matrix_eigs("left")
  np.testing.assert_allclose(eta, g[f"{tag}_eta"], rtol=max(rtol, 1e-9))
  lh = np.asarray(l)
  assert lh.dtype == dtype                          # a real state keeps a real eigenvector
  np.testing.assert_allclose(np.asarray(imps.unit_cell_transfer_operator("left", l)), eta * lh,
                             rtol=1e-6, atol=1e-7 * abs(eta) * np.abs(lh).max())
  lam_norm = imps.canonicalize()
  np.testing.assert_allclose(lam_norm, float(g[f"{tag}_lam_norm"]), rtol=1e-8)
  schmidt = np.sort(np.abs(1.0 / np.diag(np.asarray(imps.connector_matrix))))[::-1]
  np.testing.assert_allclose(schmidt, g[f"{tag}_schmidt"], rtol=1e-7, atol=1e-10)
  assert imps.center_position == n - 1 and imps.tensors[0].dtype == dtype
  assert imps.check_canonical() < 1e-10
  assert imps.check_orthonormality("l", n - 1) < 1e-8   # last tensor x connector is a left isometry too


# ------------------------------------------------------------------ Tensor / functional API
def check_tensor_api(be, rtol):
  """`Tensor` operators and the functional API (tensor.py, linalg/operations.py, linalg/linalg.py,
  linalg/initialization.py, linalg/krylov.py of the reference; its tests: tensor_test.py,
  linalg/tests/*_test.py) on backend `be`, values against NumPy."""
  import pytest
  from tensornetwork_amd import tensor as tt, linalg as tl
  rng = np.random.default_rng(123)
  H = lambda t: np.asarray(t.array)
  close = lambda got, want: np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol)
  xa, xb = rng.standard_normal((3, 4, 5)), rng.standard_normal((5, 4, 2))
  a, b = tt.Tensor(xa, backend=be), tt.Tensor(xb, backend=be)
  assert a.shape == (3, 4, 5) and a.ndim == 3 and a.size == 60 and a.dtype == np.float64 and a.backend is be
  close(H(tl.tensordot(a, b, [[2, 1], [0, 1]])), np.tensordot(xa, xb, [[2, 1], [0, 1]]))
  close(H(a.T), xa.T); close(H(a.transpose((1, 0, 2))), xa.transpose(1, 0, 2))
  close(H(a.reshape((12, 5))), xa.reshape(12, 5)); close(H(a.ravel()), xa.ravel()); close(H(a.flatten()), xa.ravel())
  assert tt.Tensor(np.ones((1, 3, 1)), backend=be).squeeze().shape == (3,)
  close(H(a * 2.0 + 1.0 - a / 4.0), xa * 2 + 1 - xa / 4); close(H(3.0 - a), 3 - xa); close(H(2.0 * a), 2 * xa)
  close(H(a + a), 2 * xa); close(H(a - a), 0 * xa); close(H(a * a), xa * xa); close(H(a / (a * a + 1.0)), xa / (xa * xa + 1))
  m1, m2 = rng.standard_normal((4, 6)), rng.standard_normal((6, 3))
  close(H(tt.Tensor(m1, backend=be) @ tt.Tensor(m2, backend=be)), m1 @ m2)
  z = rng.standard_normal((3, 4)) + 1j * rng.standard_normal((3, 4))
  zt = tt.Tensor(z, backend=be)
  close(H(zt.conj()), z.conj()); close(H(zt.H), z.conj().T); close(H(zt.hconj()), z.conj().T)
  close(H(tl.hconj(zt, (0, 1))), z.conj()); close(H(tl.conj(zt)), z.conj())
  c = a.copy()
  assert c.array is not a.array
  close(H(c), xa)
  # ncon builder syntax: A(labels) @ B(labels)
  close(H(tt.finalize(a(-1, 1, 2) @ b(2, 1, -2))), np.einsum("abc,cbd->ad", xa, xb))
  close(H(tl.einsum("abc,cbd->ad", a, b, optimize=True)), np.einsum("abc,cbd->ad", xa, xb))
  # operations
  assert tl.shape(a) == (3, 4, 5)
  close(H(tl.reshape(a, (3, 20))), xa.reshape(3, 20)); close(H(tl.transpose(a)), xa.T)
  close(H(tl.take_slice(a, (1, 0, 2), (2, 3, 2))), xa[1:3, 0:3, 2:4])
  close(H(tl.outer(tt.Tensor(m1, backend=be), tt.Tensor(m2, backend=be))), np.multiply.outer(m1, m2))
  pos = np.abs(xa) + 0.5
  p = tt.Tensor(pos, backend=be)
  for fn, ref in ((tl.sqrt, np.sqrt), (tl.log, np.log), (tl.exp, np.exp), (tl.sin, np.sin), (tl.cos, np.cos)):
    close(H(fn(p)), ref(pos))
  close(H(tl.sign(a)), np.sign(xa)); close(H(tl.abs(a)), np.abs(xa))
  sq = rng.standard_normal((2, 5, 5))
  s = tt.Tensor(sq, backend=be)
  close(H(tl.trace(s)), np.trace(sq, axis1=-2, axis2=-1)); close(H(tl.diagonal(s)), np.diagonal(sq, axis1=-2, axis2=-1))
  close(H(tl.diagflat(tt.Tensor(np.arange(4.0), backend=be))), np.diagflat(np.arange(4.0)))
  assert tl.pivot(a, 1).shape == (3, 20) and tl.pivot(a).shape == (12, 5)
  ka, kb = rng.standard_normal((2, 3, 4, 5)), rng.standard_normal((6, 7))
  kr = tl.kron(tt.Tensor(ka, backend=be), tt.Tensor(kb, backend=be))
  assert kr.shape == (2, 3, 6, 4, 5, 7)
  close(H(kr).reshape(36, 140), np.kron(ka.reshape(6, 20), kb))
  with pytest.raises(ValueError, match="even number of legs"):
    tl.kron(a, tt.Tensor(kb, backend=be))
  # decompositions
  u, sv, vh, rest = tl.svd(a, 1)
  close(np.einsum("ik,k,kbc->ibc", H(u), H(sv), H(vh)), xa)
  assert rest.shape == (0,)
  q, r = tl.qr(a, 2)
  close(np.tensordot(H(q), H(r), 1), xa)
  r2, q2 = tl.rq(a, 1)
  close(np.tensordot(H(r2), H(q2), 1), xa)
  sym = rng.standard_normal((6, 6)); sym = sym + sym.T
  w, v = tl.eigh(tt.Tensor(sym, backend=be))
  close(H(v) @ np.diag(H(w)) @ H(v).T, sym)
  close(float(np.asarray(tl.norm(a))), np.linalg.norm(xa))
  well = rng.standard_normal((5, 5)) + 5 * np.eye(5)
  close(H(tl.inv(tt.Tensor(well, backend=be))) @ well, np.eye(5))
  import scipy.linalg
  close(H(tl.expm(tt.Tensor(sym / 10, backend=be))), scipy.linalg.expm(sym / 10))
  # initialisation
  assert H(tl.eye(3, backend=be)).tolist() == np.eye(3).tolist() and tl.eye(2, M=4, dtype=np.float32, backend=be).shape == (2, 4)
  assert tl.zeros((2, 3), dtype=np.float32, backend=be).dtype == np.float32 and float(H(tl.ones((2, 2), backend=be)).sum()) == 4.0
  assert tl.ones_like(a).shape == a.shape and tl.zeros_like(np.ones((2, 2), dtype=np.float32), backend=be).dtype == np.float32
  r1, r2_ = tl.randn((4, 3), dtype=np.float64, seed=10, backend=be), tl.randn((4, 3), dtype=np.float64, seed=10, backend=be)
  close(H(r1), H(r2_))
  ru = H(tl.random_uniform((50,), boundaries=(-2.0, -1.0), seed=3, backend=be))
  assert ru.min() >= -2.0 and ru.max() <= -1.0
  # Krylov on Tensors
  hd = tt.Tensor(sym, backend=be)
  mv = lambda x, mat: tl.tensordot(mat, x, 1)
  vals, vecs = tl.eigsh_lanczos(mv, args=[hd], x0=tt.Tensor(rng.standard_normal(6), backend=be), num_krylov_vecs=6,
                                numeig=1)
  close(vals[0], np.linalg.eigvalsh(sym)[0]); assert isinstance(vecs[0], tt.Tensor)
  x, info = tl.gmres(mv, tt.Tensor(rng.standard_normal(5), backend=be), A_args=[tt.Tensor(well, backend=be)], tol=1e-12,
                     num_krylov_vectors=5, maxiter=5)
  assert info == 0 and isinstance(x, tt.Tensor)
  with pytest.raises(ValueError, match="One of backend or x0 must be specified"):
    tl.eigsh_lanczos(mv)


# --------------------------------------------------------------------------- the boundary without the reference
def signature_mismatches(backend_cls):
  """Every public method of the reference's AbstractBackend / NumPyBackend (tests/golden/abstract_backend_signatures.json,
  a snapshot written by tests/golden/make_golden_signatures.py) must exist on `backend_cls` with the same parameter
  names in the same order and the same defaults; extra parameters are allowed only with defaults.  Where the two
  reference classes disagree (the abstract class declares `dtype` without a default for eye / ones / zeros, gives
  eigs numeig = 1) the NumPy backend -- the oracle of SURVEY 8c -- wins."""
  import inspect, json, os   # pylint: disable=import-outside-toplevel,multiple-imports
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abstract_backend_signatures.json")) as f:
    rec = json.load(f)
  want = dict(rec["AbstractBackend"])
  want.update(rec["NumPyBackend"])
  bad = []
  for name, params in sorted(want.items()):
    if name == "__init__":           # construction is the factory's business (backend_factory.py:22-46), not the interface's
      continue
    fn = getattr(backend_cls, name, None)
    if fn is None:
      bad.append(f"{name}: missing")
      continue
    mine = list(inspect.signature(fn).parameters.values())
    for i, p in enumerate(params):
      if i >= len(mine) or mine[i].name != p["name"] or mine[i].kind.name != p["kind"]:
        got = f"{mine[i].name} ({mine[i].kind.name})" if i < len(mine) else None
        bad.append(f"{name}: parameter {i} is {got}, the reference has {p['name']} ({p['kind']})")
        break
      has = mine[i].default is not inspect.Parameter.empty
      if has != ("default" in p) or (has and repr(mine[i].default) != p["default"]):
        bad.append(f"{name}: default of {p['name']!r} is {mine[i].default!r}, the reference has {p.get('default', '<none>')}")
        break
    for e in mine[len(params):]:
      if e.default is inspect.Parameter.empty and e.kind.name not in ("VAR_POSITIONAL", "VAR_KEYWORD"):
        bad.append(f"{name}: extra required parameter {e.name!r}")
  return bad


def run_high_rank_cases(be):
  """Tensors of more than 16 axes (TNH_MAX_RANK): transpose (coalesced to one launch, and a reversal that needs two
  passes), broadcast arithmetic, slices, and a tensordot over scattered axes -- bit for bit against NumPy."""
  rng = np.random.default_rng(21)
  x = rng.standard_normal((2,) * 18).astype(np.float32)
  d = be.convert_to_tensor(x)
  for perm in ([17] + list(range(17)), list(range(9, 18)) + list(range(9)), list(range(17, -1, -1)),
               [int(p) for p in rng.permutation(18)]):
    np.testing.assert_array_equal(np.asarray(be.transpose(d, perm)), np.transpose(x, perm), err_msg=str(perm))
  y = rng.standard_normal((2,) * 18).astype(np.float32)
  np.testing.assert_array_equal(np.asarray(be.addition(d, be.convert_to_tensor(y))), x + y)
  row = rng.standard_normal((2,) * 3).astype(np.float32)                     # broadcast over the 15 leading axes
  np.testing.assert_array_equal(np.asarray(be.multiply(d, be.convert_to_tensor(row))), x * row)
  np.testing.assert_array_equal(np.asarray(be.slice(d, (0,) * 17 + (1,), (2,) * 17 + (1,))), x[..., 1:2])
  a = rng.standard_normal((2,) * 17).astype(np.float64)
  b = rng.standard_normal((2,) * 17).astype(np.float64)
  axes = [[1, 5, 16, 9], [0, 7, 3, 11]]
  got = np.asarray(be.tensordot(be.convert_to_tensor(a), be.convert_to_tensor(b), axes))
  assert got.shape == (2,) * 26
  np.testing.assert_allclose(got, np.tensordot(a, b, axes), rtol=1e-12, atol=1e-12)


def run_index_update_tensor_cases(be, dtypes=(np.float32, np.float64, np.complex64, np.complex128)):
  """index_update with a TENSOR assignee = NumPy's boolean-mask assignment (numpy_backend.py:548-552:
  t = copy(tensor); t[mask] = assignee), bit for bit; masks with no / all / scattered set entries, sizes that are
  not multiples of the kernels' 1024-element blocks, and the ValueError of a count mismatch."""
  import pytest  # pylint: disable=import-outside-toplevel
  rng = np.random.default_rng(31)
  for dt in dtypes:
    for shape, density in [((7,), 0.5), ((33, 65), 0.3), ((5, 6, 70), 0.9), ((4097,), 0.01), ((3, 1024), 1.0), ((50, 50), 0.0)]:
      x = rng.standard_normal(shape).astype(dt)
      if np.dtype(dt).kind == "c":
        x = (x + 1j * rng.standard_normal(shape)).astype(dt)
      mask = rng.random(shape) < density
      vals = (np.arange(int(mask.sum())) + 100).astype(dt)
      ref = np.copy(x)
      ref[mask] = vals
      got = be.index_update(be.convert_to_tensor(x), mask, be.convert_to_tensor(vals) if vals.size != 1 else vals)
      np.testing.assert_array_equal(np.asarray(got), ref, err_msg=f"{np.dtype(dt).name} {shape} {density}")
    x = rng.standard_normal((10, 10)).astype(dt)
    mask = x.real > 0
    with pytest.raises(ValueError):
      be.index_update(be.convert_to_tensor(x), mask, be.convert_to_tensor(np.zeros(int(mask.sum()) + 3, dtype=dt)))
