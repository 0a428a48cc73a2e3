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
