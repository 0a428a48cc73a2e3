"""Does a power-of-two row stride hurt the long-K bf16 GEMM?   (GPU box only; prepared for the next round)

  python tools/ld_pad_probe.py [--m 8192 --n 8192 --k 65536]

Times the ping-pong kernel on the same M x N x K problem with operands stored at leading dimension
K (row stride 2^17 B at K = 65536) and at K + pad for a few pads, zero-filled and random.  If padded
strides recover the short-K rate (1.95 PF zero-filled at K = 8192 vs 1.64 PF at K = 65536, DESIGN.md
section 4), the tensordot lowering can give its permuted copies a padded leading dimension for free."""
import argparse, ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=8192)
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--k", type=int, default=65536)
ap.add_argument("--pads", default="0,64,128,576,4160")
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
be = ta.get_hip_backend()
m, n, k = a.m, a.n, a.k
for fill in ("zeros", "random"):
  for pad in [int(p) for p in a.pads.split(",")]:
    ld = k + pad                                     # elements; pads are multiples of 8 (16-B rows)
    if fill == "zeros":
      A, B = be.zeros((m * ld,), dtype=ta.bfloat16), be.zeros((n * ld,), dtype=ta.bfloat16)
    else:
      A = be.device_random((m * ld,), dtype=ta.bfloat16, seed=1, normal=False, a=-1.0, b=1.0)
      B = be.device_random((n * ld,), dtype=ta.bfloat16, seed=2, normal=False, a=-1.0, b=1.0)
    C = DeviceTensor.empty((m, n), _lib.BF16)

    def call():
      _lib.check(be.lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, m, n, k, ctypes.c_void_p(A.ptr), ld,
                                 ctypes.c_void_p(B.ptr), ld, ctypes.c_void_p(C.ptr), n, 1, 0, 0, 0))
    call(); call()
    s = _lib.Event().record()
    for _ in range(a.iters):
      call()
    e = _lib.Event().record()
    e.synchronize()
    ms = s.elapsed_ms(e) / a.iters
    print(json.dumps({"m": m, "n": n, "k": k, "ld": ld, "row_stride_bytes": 2 * ld, "fill": fill, "ms": ms,
                      "tflops": 2.0 * m * n * k / ms / 1e9, "kernel": be.lib.tnh_gemm_last_kernel().decode()}),
          flush=True)
    del A, B, C
