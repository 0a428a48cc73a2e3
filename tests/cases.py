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
