"""pytest configuration: markers, import paths and the golden-fixture loader."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, HERE):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


class Golden:
  """tests/golden/golden.npz + cases.json (made by tests/golden/make_golden.py)."""

  def __init__(self):
    self.arrays = np.load(os.path.join(HERE, "golden", "golden.npz"))
    with open(os.path.join(HERE, "golden", "cases.json")) as f:
      self.cases = json.load(f)

  def __getitem__(self, name):
    return self.arrays[name]


@pytest.fixture(scope="session")
def golden():
  return Golden()


class GoldenLinalg(Golden):
  """tests/golden/golden_linalg.npz + cases_linalg.json (made by make_golden_linalg.py)."""

  def __init__(self):  # pylint: disable=super-init-not-called
    self.arrays = np.load(os.path.join(HERE, "golden", "golden_linalg.npz"))
    with open(os.path.join(HERE, "golden", "cases_linalg.json")) as f:
      self.cases = json.load(f)


@pytest.fixture(scope="session")
def golden_linalg():
  return GoldenLinalg()


class GoldenComplex(Golden):
  """tests/golden/golden_complex.npz + cases_complex.json (made by make_golden_complex.py)."""

  def __init__(self):  # pylint: disable=super-init-not-called
    self.arrays = np.load(os.path.join(HERE, "golden", "golden_complex.npz"))
    with open(os.path.join(HERE, "golden", "cases_complex.json")) as f:
      self.cases = json.load(f)


@pytest.fixture(scope="session")
def golden_complex():
  return GoldenComplex()


@pytest.fixture(scope="session")
def hip():
  """The hip backend bound to cuda:0 -- fails loudly if libtnhip or the GPU is missing."""
  import tensornetwork_amd as ta
  be = ta.get_hip_backend()
  be.lib  # initialise: raises HipRuntimeError without a gfx950 device  # pylint: disable=pointless-statement
  return be
