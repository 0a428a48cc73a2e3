"""Host-only dry run of the MERA layer energy at chi (no GPU, no data): which contractions are read in place by the
view GEMM, which operands still go through a K1 permute, and how many bytes those move.
  python tools/mera_trace.py --chi 32"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib, hip_backend, device_tensor, workloads, contractors

LOG = []


class FakeTensor(device_tensor.DeviceTensor):
  def __init__(self, shape, code, alias=None):
    self._shape = tuple(int(s) for s in shape); self._code = code; self._block = None; self._offset = 0; self._alias = None; self._pad = None
  @classmethod
  def empty(cls, shape, code, alias=None):
    return cls(shape, code)
  def _clone(self, shape, pad=None):
    t = FakeTensor(shape, self._code)
    t._pad = pad
    return t
  @property
  def ptr(self):
    return 0


device_tensor.DeviceTensor.empty = FakeTensor.empty
hip_backend.DeviceTensor.empty = FakeTensor.empty
device_tensor.DeviceTensor._fresh = classmethod(lambda cls, shape, code, nbytes, alias=None: FakeTensor(shape, code))


class FakeLib:
  def tnh_gemm_view(self, code, out_code, m, n, k, a, va, b, vb, out, ldc):
    LOG.append(("view_gemm", m, n, k, (va._obj.sk0, va._obj.sr1, va._obj.sk1), (vb._obj.sk0, vb._obj.sr1, vb._obj.sk1)))
    return 0


class TraceBackend(hip_backend.HipBackend):
  @property
  def lib(self):
    return FakeLib()
  def cast(self, tensor, dtype):
    return tensor
  def transpose(self, tensor, perm=None):
    nd = tensor.ndim
    perm = tuple(range(nd - 1, -1, -1)) if perm is None else tuple(perm)
    if perm == tuple(range(nd)):
      return tensor
    LOG.append(("permute", tensor.shape, perm, tensor.size * 2))
    return FakeTensor([tensor.shape[p] for p in perm], tensor.code)
  def _gemm(self, a, b, trans_a, trans_b, m, n, k, lda, ldb, batch=1, stride_a=0, stride_b=0, out_shape=None, out_code=None,
            alias=None):
    LOG.append(("gemm", int(trans_a), int(trans_b), m, n, k, batch))
    return FakeTensor(out_shape if out_shape is not None else (m, n), a.code)
  def _dense(self, tensor):
    if tensor.pad is None:
      return tensor
    LOG.append(("permute", tensor.shape, "dense copy of a padded result", tensor.size * 2))
    return FakeTensor(tensor.shape, tensor.code)
  def _outer(self, a, b, out_shape, alias=None):
    return FakeTensor(out_shape, a.code)
  def _strided_copy(self, tensor, shape, strides, offset):
    return FakeTensor(shape, tensor.code)
  def trace(self, tensor, *a, **k):
    return FakeTensor((), tensor.code)
  def addition(self, a, b):
    return a
  def multiply(self, a, b):
    return a


ap = argparse.ArgumentParser()
ap.add_argument("--chi", type=int, default=32)
ap.add_argument("--pad", action="store_true", help="row-padded results on (HipBackend.pad_results)")
a = ap.parse_args()
be = TraceBackend()
be.pad_results = a.pad
chi = a.chi
ham = FakeTensor((chi,) * 6, _lib.BF16)
state = FakeTensor((chi,) * 6, _lib.BF16)
iso = FakeTensor((chi, chi, chi), _lib.BF16)
dis = FakeTensor((chi,) * 4, _lib.BF16)
for placement in ("left", "right"):
  mark = len(LOG)
  nodes = workloads.mera_layer_network(be, ham, state, iso, dis, placement)
  try:
    contractors.branch(nodes, nbranch=2)
  except Exception as exc:  # pylint: disable=broad-except
    print("trace stopped:", type(exc).__name__, exc)
  print(f"== placement {placement}")
  for rec in LOG[mark:]:
    if rec[0] == "permute" and rec[3] > 1e8:
      print(f"  permute {rec[1]} -> {rec[2]}  {2 * rec[3] / 1e9:.1f} GB moved")
    elif rec[0] == "view_gemm":
      print(f"  view GEMM M={rec[1]} N={rec[2]} K={rec[3]} a(sk0,sr1,sk1)={rec[4]} b={rec[5]}  {2.0 * rec[1] * rec[2] * rec[3] / 1e12:.1f} TFLOP")
    elif rec[0] == "gemm" and 2.0 * rec[3] * rec[4] * rec[5] > 1e11:
      print(f"  gemm tA={rec[1]} tB={rec[2]} M={rec[3]} N={rec[4]} K={rec[5]}  {2.0 * rec[3] * rec[4] * rec[5] / 1e12:.2f} TFLOP")
