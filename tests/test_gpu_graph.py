"""Graph surgery on device-resident networks: flatten / split edges reshape and permute tensors in
HBM (K1 permute kernels + metadata reshapes), reduced_density conjugates on the device.  Same
checker as the CPU suite (tests/cases.py:check_graph_surgery), restating the reference's
network_test.py / tensornetwork_test.py / network_operations_test.py cases."""
import numpy as np
import pytest

import tensornetwork_amd as ta
import cases
from oracle.numpy_oracle import OracleBackend

pytestmark = pytest.mark.gpu


def test_graph_surgery_reference_cases_on_device(hip):
  cases.check_graph_surgery(hip, 1e-12)


def test_switch_backend_host_network_into_hbm(hip):
  """network_operations_test.py:376-411: build on a host backend, switch, contract on the device."""
  rng = np.random.default_rng(5)
  ta_, tb_ = rng.standard_normal((4, 5, 6)), rng.standard_normal((6, 5, 3))
  a, b = ta.Node(ta_, backend=OracleBackend()), ta.Node(tb_, backend=OracleBackend())
  ta.connect(a[2], b[0]); ta.connect(a[1], b[1])
  ta.switch_backend([a, b], hip)
  assert a.backend is hip and isinstance(a.tensor, ta.DeviceTensor) and isinstance(b.tensor, ta.DeviceTensor)
  out = ta.contract_between(a, b)
  np.testing.assert_allclose(np.asarray(out.tensor), np.einsum("abc,cbd->ad", ta_, tb_), rtol=1e-12)
