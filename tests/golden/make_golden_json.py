"""Wire-format fixture: a small network serialised by the REFERENCE's
``tn.nodes_to_json`` (NumPy backend), plus the check that the reference reads back what
``tensornetwork_amd.nodes_to_json`` writes.  Run in the build container::

    python tests/golden/make_golden_json.py

Writes tests/golden/network_ref.json (committed)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, "_stubs"), "/root/reference", REPO]

import tensornetwork as tn  # noqa: E402  pylint: disable=wrong-import-position

rng = np.random.default_rng(41)
a = tn.Node(rng.standard_normal((2, 3, 4)), name="a", axis_names=["x", "y", "z"], backend="numpy")
b = tn.Node(rng.standard_normal((4, 3, 5)).astype(np.float32), name="b", backend="numpy")
c = tn.Node(rng.standard_normal((5, 2)) + 1j * rng.standard_normal((5, 2)), name="c", backend="numpy")
e1 = tn.connect(a[2], b[0], name="ab")
e2 = tn.connect(a[1], b[1], name="ab2")
e3 = tn.connect(b[2], c[0], name="bc")
text = tn.nodes_to_json([a, b, c], edge_binding={"bond": e1, "pair": [e2, e3], "open": a[0]})
result = tn.contractors.greedy([a, b, c], output_edge_order=[a[0], c[1]]).tensor
with open(os.path.join(HERE, "network_ref.json"), "w") as f:
  json.dump({"network": text, "result_re": np.real(result).tolist(), "result_im": np.imag(result).tolist()}, f)

# reverse direction: the reference reads what this library writes
import tensornetwork_amd as ta  # noqa: E402  pylint: disable=wrong-import-position
from oracle import numpy_oracle as orc  # noqa: E402  pylint: disable=wrong-import-position


class NamedNumpy(orc.OracleBackend):
  name = "numpy"


be = NamedNumpy()
nodes, binding = ta.nodes_from_json(text, backend=be)
text2 = ta.nodes_to_json(nodes, edge_binding={k: list(v) for k, v in binding.items()})
nodes3, binding3 = tn.nodes_from_json(text2)
out = tn.contractors.greedy(nodes3, output_edge_order=[binding3["open"][0], nodes3[2][1]]).tensor
np.testing.assert_allclose(out, result, rtol=1e-6)
assert set(binding3) == {"bond", "pair", "open"} and len(binding3["pair"]) == 2
print("ok: reference reads tensornetwork_amd's JSON; wrote network_ref.json", len(text), "bytes")
