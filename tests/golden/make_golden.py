"""Generate golden input/output fixtures by running the REFERENCE itself.

Run in the build container only (``/root/reference`` does not exist on the GPU
box)::

    python tests/golden/make_golden.py

It imports google/TensorNetwork v0.4.6 from /root/reference with inert stubs for
the absent h5py / graphviz (and ``opt_einsum`` -> tensornetwork_amd.pathfinder),
drives its NumPy backend (``backends/numpy/numpy_backend.py``) through
``tn.ncon``, ``tn.contract_between``, ``tn.contract``, ``tn.split_node`` and the
contractors, and stores inputs + outputs in ``tests/golden/golden.npz`` with the
case descriptions in ``tests/golden/cases.json``.  The fixtures are committed;
the tests only read them.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, "_stubs"), "/root/reference", REPO]

import tensornetwork as tn  # noqa: E402  pylint: disable=wrong-import-position
from tensornetwork.backends.numpy import numpy_backend  # noqa: E402  pylint: disable=wrong-import-position

assert tn.__version__ == "0.4.6", tn.__version__
BE = numpy_backend.NumPyBackend()
ARR = {}
CASES = {"reference_version": tn.__version__, "numpy_version": np.__version__}


def put(name, array):
  assert name not in ARR, name
  ARR[name] = np.asarray(array)
  return name


def rnd(rng, shape, dtype):
  x = rng.standard_normal(shape)
  if np.dtype(dtype).kind == "c":
    x = x + 1j * rng.standard_normal(shape)
  return x.astype(dtype)


# ---------------------------------------------------------------- tensordot
def gen_tensordot():
  rng = np.random.default_rng(11)
  cases = []
  specs = [
      # (shape_a, shape_b, axes)
      ((2, 3, 4), (2, 3, 4), [[1, 2], [1, 2]]),           # numpy_backend_test.py:12-18
      ((2, 3, 4), (4, 5), 1),                              # int axes
      ((3, 4), (4, 3), 2 if False else 1),
      ((2, 3), (4,), 0),                                   # outer product
      ((4, 3, 2), (2, 3, 4), [[0, 1, 2], [2, 1, 0]]),      # full contraction, permuted
      ((3, 3, 3), (3, 3, 3), [[0, 1, 2], [0, 1, 2]]),
      ((5, 4, 3, 2), (3, 5, 6), [[0, 2], [1, 0]]),         # scattered axes
      ((6, 2, 5), (5, 6, 3), [[2, 0], [0, 1]]),            # pair order differs from memory order
      ((4, 6, 5), (6, 4, 7), [[0, 1], [1, 0]]),            # KM form on a
      ((7, 4, 6), (3, 6, 4), [[1, 2], [2, 1]]),            # NK form on b, swapped pairs
      ((8, 9), (9, 10), [[1], [0]]),
      ((8, 9), (10, 9), [[1], [1]]),
      ((9, 8), (9, 10), [[0], [0]]),
      ((9, 8), (10, 9), [[0], [1]]),
      ((2, 3, 4, 5), (3, 5, 4, 2), [[0, 3, 1], [3, 1, 0]]),  # tensornetwork_test.py:499-523 layout
      ((5,), (5,), 1),
      ((1, 5, 1), (5, 1), [[1], [0]]),
      ((33, 17), (17, 65), 1),
      ((64, 48, 2), (2, 48, 40), [[1, 2], [1, 0]]),
  ]
  for n, (sa, sb, axes) in enumerate(specs):
    for dt in ("float32", "float64", "complex128"):
      a, b = rnd(rng, sa, dt), rnd(rng, sb, dt)
      out = BE.tensordot(a, b, axes)
      key = f"td{n}_{dt}"
      cases.append({"a": put(key + "_a", a), "b": put(key + "_b", b), "axes": axes,
                    "out": put(key + "_out", out)})
  # the reference's own KAT (numpy_backend_test.py:12-18): 2*ones . ones -> 24
  a = 2 * np.ones((2, 3, 4))
  b = np.ones((2, 3, 4))
  cases.append({"a": put("td_kat_a", a), "b": put("td_kat_b", b), "axes": [[1, 2], [1, 2]],
                "out": put("td_kat_out", BE.tensordot(a, b, ((1, 2), (1, 2))))})
  CASES["tensordot"] = cases


# ------------------------------------------------------------------ transpose
def gen_transpose():
  rng = np.random.default_rng(12)
  cases = []
  specs = [((2, 3, 4), None), ((2, 3, 4), (2, 0, 1)), ((5, 1, 7), (2, 1, 0)), ((4, 5, 6, 7), (1, 3, 0, 2)),
           ((3, 4, 5, 6), (0, 1, 3, 2)), ((2, 2, 2, 2, 2, 2), (5, 3, 1, 4, 2, 0)), ((70, 33), (1, 0)),
           ((17, 2, 19), (2, 1, 0)), ((16, 16, 16), (1, 0, 2)), ((3, 65, 2, 66), (3, 2, 1, 0))]
  for n, (shape, perm) in enumerate(specs):
    for dt in ("float32", "float64", "int64", "complex128", "float16"):
      x = rng.integers(-1000, 1000, size=shape).astype(dt)
      out = np.ascontiguousarray(BE.transpose(x, perm))
      cases.append({"x": put(f"tr{n}_{dt}_x", x), "perm": perm, "out": put(f"tr{n}_{dt}_out", out)})
  CASES["transpose"] = cases


# ----------------------------------------------------------------------- ncon
def gen_ncon():
  rng = np.random.default_rng(13)
  cases = []

  def add(name, tensors, structure, con_order=None, out_order=None):
    out = tn.ncon(tensors, structure, con_order=con_order, out_order=out_order, backend="numpy")
    cases.append({"tensors": [put(f"nc_{name}_t{i}", t) for i, t in enumerate(tensors)],
                  "structure": structure, "con_order": con_order, "out_order": out_order,
                  "out": put(f"nc_{name}_out", out)})

  for dt in ("float32", "float64"):
    a, b = rnd(rng, (10, 10), dt), rnd(rng, (10, 10), dt)
    add(f"matmul_{dt}", [a, b], [[-1, 1], [1, -2]])                    # README / config 1
    add(f"matmulT_{dt}", [a, b], [[-2, 1], [1, -1]])
    add(f"order_{dt}", [rnd(rng, (3, 4, 5), dt), rnd(rng, (5, 3, 6), dt), rnd(rng, (4, 6), dt)],
        [[1, 2, 3], [3, 1, 4], [2, 4]], con_order=[2, 4, 1, 3])
    add(f"outorder_{dt}", [rnd(rng, (3, 4, 5), dt), rnd(rng, (5, 6), dt)], [[-1, -2, 1], [1, -3]],
        out_order=[-3, -1, -2])
    add(f"trace_{dt}", [rnd(rng, (4, 4), dt)], [[1, 1]])
    add(f"ptrace_{dt}", [rnd(rng, (3, 4, 3, 5), dt), rnd(rng, (5, 4), dt)], [[1, 2, 1, 3], [3, 2]])
    add(f"ptrace_open_{dt}", [rnd(rng, (3, 4, 3, 5), dt)], [[1, -1, 1, -2]])
    add(f"outer_{dt}", [rnd(rng, (3,), dt), rnd(rng, (4, 2), dt)], [[-1], [-2, -3]])
    add(f"three_{dt}", [rnd(rng, (2, 3, 4, 5), dt), rnd(rng, (4, 6, 7), dt), rnd(rng, (5, 6, 8), dt)],
        [[-1, -2, 1, 2], [1, 3, -3], [2, 3, -4]])
    add(f"batch_{dt}", [rnd(rng, (6, 3, 4), dt), rnd(rng, (6, 4, 5), dt)], [[-1, -2, 1], [-1, 1, -3]])
    add(f"str_{dt}", [rnd(rng, (3, 4), dt), rnd(rng, (4, 5), dt)], [["-a", "x"], ["x", "-b"]])
  # binary MERA ascending-superoperator network (examples/custom_path_solvers/example.py:40-49)
  for chi in (2, 3):
    u = rng.random((chi, chi, chi, chi))
    w = rng.random((chi, chi, chi))
    ham = rng.random((chi,) * 6)
    tensors = [u, u, w, w, w, ham, u, u, w, w, w]
    connects = [[1, 3, 10, 11], [4, 7, 12, 13], [8, 10, -4], [11, 12, -5], [13, 14, -6],
                [2, 5, 6, 3, 4, 7], [1, 2, 9, 17], [5, 6, 16, 15], [8, 9, -1], [17, 16, -2], [15, 14, -3]]
    add(f"mera_chi{chi}", tensors, connects)
  CASES["ncon"] = cases


# ---------------------------------------------------------- contract_between
def gen_contract_between():
  rng = np.random.default_rng(14)
  cases = []
  for dt in ("float32", "float64"):
    # tensornetwork_test.py:499-523: three shared edges, output order [b[2], a[2]]
    a_val, b_val = rnd(rng, (2, 3, 4, 5), dt), rnd(rng, (3, 5, 4, 2), dt)
    a, b = tn.Node(a_val, backend="numpy"), tn.Node(b_val, backend="numpy")
    tn.connect(a[0], b[3]); tn.connect(b[1], a[3]); tn.connect(a[1], b[0])
    d = tn.contract_between(a, b, output_edge_order=[b[2], a[2]])
    cases.append({"kind": "between", "a": put(f"cb_{dt}_a", a_val), "b": put(f"cb_{dt}_b", b_val),
                  "connect": [[0, 3], [3, 1], [1, 0]], "order": [["b", 2], ["a", 2]],
                  "out": put(f"cb_{dt}_out", d.tensor)})
    # config-2 layouts at D=6: L0 a[2]^b[0], a[3]^b[1]; L1 a[1]^b[2], a[3]^b[0]
    for lname, conn in (("L0", [[2, 0], [3, 1]]), ("L1", [[1, 2], [3, 0]])):
      a_val, b_val = rnd(rng, (6, 6, 6, 6), dt), rnd(rng, (6, 6, 6, 6), dt)
      a, b = tn.Node(a_val, backend="numpy"), tn.Node(b_val, backend="numpy")
      for x, y in conn:
        tn.connect(a[x], b[y])
      d = tn.contract_between(a, b)
      cases.append({"kind": "between", "a": put(f"cb_{lname}_{dt}_a", a_val), "b": put(f"cb_{lname}_{dt}_b", b_val),
                    "connect": conn, "order": None, "out": put(f"cb_{lname}_{dt}_out", d.tensor)})
    # tensornetwork_test.py:190-214 "real physics": edge-at-a-time with a final trace
    a_val, b_val, c_val = rnd(rng, (2, 3, 4, 5), dt), rnd(rng, (4, 6, 7), dt), rnd(rng, (5, 6, 8), dt)
    a, b, c = (tn.Node(v, backend="numpy") for v in (a_val, b_val, c_val))
    e1 = tn.connect(a[2], b[0]); e2 = tn.connect(c[0], a[3]); e3 = tn.connect(b[1], c[1])
    tn.contract(e1); tn.contract(e2)
    val = tn.contract(e3)
    cases.append({"kind": "physics", "a": put(f"ph_{dt}_a", a_val), "b": put(f"ph_{dt}_b", b_val),
                  "c": put(f"ph_{dt}_c", c_val), "out": put(f"ph_{dt}_out", val.tensor)})
    # trace edge: tensornetwork_test.py:526-533
    t_val = rnd(rng, (3, 4, 3), dt)
    t = tn.Node(t_val, backend="numpy")
    tn.connect(t[0], t[2])
    r = tn.contract_between(t, t)
    cases.append({"kind": "trace", "a": put(f"tre_{dt}_a", t_val), "out": put(f"tre_{dt}_out", r.tensor)})
    # outer product
    x_val, y_val = rnd(rng, (2, 3), dt), rnd(rng, (4,), dt)
    r = tn.outer_product(tn.Node(x_val, backend="numpy"), tn.Node(y_val, backend="numpy"))
    cases.append({"kind": "outer", "a": put(f"op_{dt}_a", x_val), "b": put(f"op_{dt}_b", y_val),
                  "out": put(f"op_{dt}_out", r.tensor)})
  CASES["contract_between"] = cases


# ------------------------------------------------------------------ split_node
def prescribed_spectrum(rng, m, n, s, dt):
  q1, _ = np.linalg.qr(rng.standard_normal((m, m)))
  q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
  r = min(m, n)
  return ((q1[:, :r] * np.asarray(s)[:r]) @ q2[:r, :]).astype(dt)


def gen_split():
  rng = np.random.default_rng(15)
  cases = []

  def add(name, val, left_axes, right_axes, **kw):
    node = tn.Node(val, backend="numpy")
    left, right, trun = tn.split_node(node, [node[i] for i in left_axes], [node[i] for i in right_axes], **kw)
    full = tn.contract_between(left, right)
    cases.append({"x": put(f"sp_{name}_x", val), "left": list(left_axes), "right": list(right_axes), "kw": kw,
                  "left_shape": list(left.shape), "right_shape": list(right.shape),
                  "trun": put(f"sp_{name}_trun", trun), "recon": put(f"sp_{name}_recon", full.tensor),
                  "s": put(f"sp_{name}_s", BE.svd(np.transpose(val, list(left_axes) + list(right_axes)),
                                                  len(left_axes), kw.get("max_singular_values"),
                                                  kw.get("max_truncation_err"),
                                                  relative=kw.get("relative", False))[1])})

  for dt in ("float32", "float64"):
    add(f"zeros_{dt}", np.zeros((2, 3, 4, 5), dtype=dt), (0, 1), (2, 3))           # split_node_test.py:22-32
    add(f"mixed_{dt}", rnd(rng, (2, 3, 4, 5), dt), (0, 2), (1, 3))                  # :36-47
    add(f"plain_{dt}", rnd(rng, (4, 5, 6), dt), (0, 1), (2,))
    add(f"tall_{dt}", rnd(rng, (30, 7), dt), (0,), (1,))
    add(f"wide_{dt}", rnd(rng, (7, 30), dt), (0,), (1,))
    # decompositions_test.py:55-66: spectrum 0..9, keep 7 -> s = 9..3, trun = 2,1,0
    spec = np.arange(9, -1, -1, dtype=np.float64)
    val = prescribed_spectrum(rng, 10, 10, spec, dt)
    add(f"spec_k7_{dt}", val, (0,), (1,), max_singular_values=7)
    add(f"spec_kbig_{dt}", val, (0,), (1,), max_singular_values=20)                # :68-77
    add(f"spec_err_{dt}", val, (0,), (1,), max_truncation_err=float(np.sqrt(5.1)))  # :79-90
    # :92-108 relative vs absolute: diag(2, 1, .2, .1), err 0.2
    dval = np.diag([2.0, 1.0, 0.2, 0.1]).astype(dt)
    add(f"abs_{dt}", dval, (0,), (1,), max_truncation_err=0.2, relative=False)
    add(f"rel_{dt}", dval, (0,), (1,), max_truncation_err=0.2, relative=True)
    add(f"rank6_{dt}", rnd(rng, (3, 3, 3, 3, 3, 3), dt), (0, 2, 4), (1, 3, 5), max_singular_values=9)
  CASES["split_node"] = cases


# ----------------------------------------------------------------- contractors
def gen_contractors():
  rng = np.random.default_rng(16)
  cases = []
  for dt in ("float32", "float64"):
    # MPS overlap <psi|psi>, 6 sites (config-4 topology, small D)
    n_sites, d, D = 6, 2, 5
    dims = [1] + [D] * (n_sites - 1) + [1]
    kets = [rnd(rng, (dims[i], d, dims[i + 1]), dt) for i in range(n_sites)]
    nodes_k = [tn.Node(k, backend="numpy") for k in kets]
    nodes_b = [tn.Node(np.conj(k), backend="numpy") for k in kets]
    for i in range(n_sites):
      tn.connect(nodes_k[i][1], nodes_b[i][1])
      if i + 1 < n_sites:
        tn.connect(nodes_k[i][2], nodes_k[i + 1][0])
        tn.connect(nodes_b[i][2], nodes_b[i + 1][0])
    tn.connect(nodes_k[0][0], nodes_b[0][0])
    tn.connect(nodes_k[-1][2], nodes_b[-1][2])
    val = tn.contractors.greedy(nodes_k + nodes_b)
    cases.append({"kind": "mps_overlap", "kets": [put(f"ct_{dt}_ket{i}", k) for i, k in enumerate(kets)],
                  "out": put(f"ct_{dt}_overlap", val.tensor)})
    # random 3-regular graph, 8 vertices (north-star topology, small)
    import networkx as nx
    g = nx.random_regular_graph(3, 8, seed=6)
    vals = {v: rnd(rng, (3, 3, 3), dt) for v in g.nodes}
    nodes = {v: tn.Node(vals[v], backend="numpy") for v in g.nodes}
    slot = {v: 0 for v in g.nodes}
    edges = []
    for x, y in sorted(g.edges):
      tn.connect(nodes[x][slot[x]], nodes[y][slot[y]])
      edges.append([int(x), slot[x], int(y), slot[y]])
      slot[x] += 1
      slot[y] += 1
    val = tn.contractors.greedy(list(nodes.values()))
    cases.append({"kind": "regular", "tensors": [put(f"ct_{dt}_rr{v}", vals[v]) for v in sorted(g.nodes)],
                  "edges": edges, "out": put(f"ct_{dt}_rr_out", val.tensor)})
    # open network with output edge order (path_contractors_node_test.py style)
    a_val, b_val, c_val = rnd(rng, (2, 3, 4), dt), rnd(rng, (4, 5, 6), dt), rnd(rng, (6, 3, 7), dt)
    a, b, c = (tn.Node(v, backend="numpy") for v in (a_val, b_val, c_val))
    tn.connect(a[2], b[0]); tn.connect(b[2], c[0]); tn.connect(a[1], c[1])
    for alg in ("greedy", "optimal", "branch", "auto"):
      an, bn, cn = (tn.Node(v, backend="numpy") for v in (a_val, b_val, c_val))
      tn.connect(an[2], bn[0]); tn.connect(bn[2], cn[0]); tn.connect(an[1], cn[1])
      val = getattr(tn.contractors, alg)([an, bn, cn], output_edge_order=[cn[2], an[0], bn[1]])
      cases.append({"kind": "open3", "alg": alg, "a": f"ct_{dt}_o3a", "b": f"ct_{dt}_o3b", "c": f"ct_{dt}_o3c",
                    "out": put(f"ct_{dt}_o3_{alg}", val.tensor)})
    put(f"ct_{dt}_o3a", a_val); put(f"ct_{dt}_o3b", b_val); put(f"ct_{dt}_o3c", c_val)
  CASES["contractors"] = cases


# -------------------------------------------------------------- misc backend ops
def gen_misc():
  rng = np.random.default_rng(17)
  cases = []
  for dt in ("float32", "float64", "complex128"):
    x = rnd(rng, (3, 4, 5), dt)
    v = rnd(rng, (5,), dt)
    w = rnd(rng, (3,), dt)
    m = rnd(rng, (2, 6, 6), dt)
    cases.append({
        "dtype": dt, "x": put(f"ms_{dt}_x", x), "v": put(f"ms_{dt}_v", v), "w": put(f"ms_{dt}_w", w),
        "m": put(f"ms_{dt}_m", m),
        "brm": put(f"ms_{dt}_brm", BE.broadcast_right_multiplication(x, v)),     # numpy_backend_test.py:750-766
        "blm": put(f"ms_{dt}_blm", BE.broadcast_left_multiplication(w, x)),      # :768-785
        "sum12": put(f"ms_{dt}_sum12", BE.sum(x, axis=(1, 2))),                  # :844-855
        "sum0": put(f"ms_{dt}_sum0", BE.sum(x, axis=(0,))),
        "sum02": put(f"ms_{dt}_sum02", BE.sum(x, axis=(0, 2))),
        "trace": put(f"ms_{dt}_trace", BE.trace(m)),                              # :921-932
        "trace1": put(f"ms_{dt}_trace1", BE.trace(m, offset=1)),
        "matmul": put(f"ms_{dt}_matmul", BE.matmul(m, m)),                        # :858-867
        "outer": put(f"ms_{dt}_outer", BE.outer_product(v, w)),                   # :122-130
        "diagflat": put(f"ms_{dt}_diagflat", BE.diagflat(v)),
        "norm": put(f"ms_{dt}_norm", BE.norm(x)),
        "sqrtabs": put(f"ms_{dt}_sqrtabs", BE.sqrt(BE.abs(x))),
        "conj": put(f"ms_{dt}_conj", BE.conj(x)),
        "add": put(f"ms_{dt}_add", BE.addition(x, x)), "sub": put(f"ms_{dt}_sub", BE.subtraction(x, v)),
        "mul": put(f"ms_{dt}_mul", BE.multiply(x, v)), "div": put(f"ms_{dt}_div", BE.divide(x, v)),
        "slice": put(f"ms_{dt}_slice", BE.slice(x, (1, 0, 2), (2, 3, 2))),
        "diagonal": put(f"ms_{dt}_diagonal", BE.diagonal(m)),
    })
  CASES["misc"] = cases


def main():
  gen_tensordot()
  gen_transpose()
  gen_ncon()
  gen_contract_between()
  gen_split()
  gen_contractors()
  gen_misc()
  np.savez_compressed(os.path.join(HERE, "golden.npz"), **ARR)
  with open(os.path.join(HERE, "cases.json"), "w") as f:
    json.dump(CASES, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
  nbytes = os.path.getsize(os.path.join(HERE, "golden.npz"))
  print(f"wrote {len(ARR)} arrays ({nbytes / 1e6:.2f} MB) and "
        f"{sum(len(v) for v in CASES.values() if isinstance(v, list))} cases")


if __name__ == "__main__":
  main()
