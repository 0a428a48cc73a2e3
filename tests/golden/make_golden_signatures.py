"""Snapshot of the reference's backend interface (build container only: /root/reference is not on the GPU box).

    python tests/golden/make_golden_signatures.py

Writes tests/golden/abstract_backend_signatures.json: for every public method of
``tensornetwork.backends.abstract_backend.AbstractBackend`` (abstract_backend.py:27-1046) and of the NumPy
backend (``backends/numpy/numpy_backend.py``, the oracle of SURVEY 8c) its parameters -- name, kind, whether it
has a default and the default's repr -- as ``inspect.signature`` reports them.  The committed file is data; the
tests (CPU and -m gpu) hold ``HipBackend`` to it: same parameter names in the same order, same defaults.
"""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, "_stubs"), "/root/reference", REPO]

import tensornetwork as tn  # noqa: E402  pylint: disable=wrong-import-position
from tensornetwork.backends import abstract_backend  # noqa: E402  pylint: disable=wrong-import-position
from tensornetwork.backends.numpy import numpy_backend  # noqa: E402  pylint: disable=wrong-import-position


def describe(cls):
  out = {}
  for name, fn in inspect.getmembers(cls, predicate=inspect.isfunction):
    if name.startswith("_") and name != "__init__":
      continue
    params = []
    for p in inspect.signature(fn).parameters.values():
      rec = {"name": p.name, "kind": p.kind.name}
      if p.default is not inspect.Parameter.empty:
        rec["default"] = repr(p.default)
      params.append(rec)
    out[name] = params
  return out


def main():
  rec = {"reference_version": tn.__version__,
         "source": "tensornetwork/backends/abstract_backend.py:27-1046, tensornetwork/backends/numpy/numpy_backend.py",
         "AbstractBackend": describe(abstract_backend.AbstractBackend),
         "NumPyBackend": describe(numpy_backend.NumPyBackend)}
  path = os.path.join(HERE, "abstract_backend_signatures.json")
  with open(path, "w") as f:
    json.dump(rec, f, indent=1, sort_keys=True)
  print(path, len(rec["AbstractBackend"]), "abstract methods,", len(rec["NumPyBackend"]), "numpy-backend methods")


if __name__ == "__main__":
  main()
