"""One tnh_gemm_view launch with arbitrary operand views (elements): r0,sr0,sr1,k0,sk0,sk1 per operand; the operands are
zero- or normal-filled flat buffers large enough for the views.  TFLOP/s per knob.
  python tools/view_probe.py --m 1024 --n 32768 --k 1048576 --va 1024,1048576,0,1048576,1,0 \\
      --vb 32,1024,33554432,1024,1,32768 --knobs auto,auto:r1,auto:g0"""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, required=True)
ap.add_argument("--n", type=int, required=True)
ap.add_argument("--k", type=int, required=True)
ap.add_argument("--va", required=True)
ap.add_argument("--vb", required=True)
ap.add_argument("--knobs", default="auto")
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--fill", default="normal")
ap.add_argument("--ldc-pad", type=int, default=0, help="elements of padding per row of C")
a = ap.parse_args()
be = ta.get_hip_backend()
M, N, K = a.m, a.n, a.k


def span(v, rows):
  r0, sr0, sr1, k0, sk0, sk1 = v
  r1 = (rows + r0 - 1) // r0 - 1
  k1 = K // k0 - 1
  return (min(r0, rows) - 1) * sr0 + r1 * sr1 + (k0 - 1) * sk0 + k1 * sk1 + 1


def make(v, rows, seed, sigma):
  n = span(v, rows)
  n = (n + 7) // 8 * 8
  if a.fill == "zeros":
    return be.zeros((n,), dtype=ta.bfloat16)
  return be.device_random((n,), dtype=ta.bfloat16, seed=seed, normal=True, a=0.0, b=sigma)


va = [int(x) for x in a.va.split(",")]
vb = [int(x) for x in a.vb.split(",")]
A = make(va, M, 1, K ** -0.5)
B = make(vb, N, 2, 1.0)
C = DeviceTensor.empty((M, N + a.ldc_pad), _lib.BF16)
ova, ovb = _lib.OperandView(*va), _lib.OperandView(*vb)
flop = 2.0 * M * N * K
for knob in a.knobs.split(","):
  _lib.check(be.lib.tnh_gemm_set_variant(knob.encode()))
  try:
    def call():
      _lib.check(be.lib.tnh_gemm_view(_lib.BF16, _lib.BF16, M, N, K, ctypes.c_void_p(A.ptr), ctypes.byref(ova),
                                      ctypes.c_void_p(B.ptr), ctypes.byref(ovb), ctypes.c_void_p(C.ptr), N + a.ldc_pad), "view")
    call()
    be.synchronize()
    s = _lib.Event().record()
    for _ in range(a.iters):
      call()
    e = _lib.Event().record()
    e.synchronize()
    ms = s.elapsed_ms(e) / a.iters
    print(json.dumps({"gemm": [M, N, K], "va": va, "vb": vb, "knob": knob, "fill": a.fill, "ldc_pad": a.ldc_pad, "ms": round(ms, 3),
                      "tflops": round(flop / ms / 1e9, 1), "kernel": be.lib.tnh_gemm_last_kernel().decode()}), flush=True)
  finally:
    _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
