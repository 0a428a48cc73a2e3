"""Launch one GEMM shape repeatedly (target of rocprofv3 PMC passes).
  python tools/gemm_one.py --variant bf16_256pp:b1 --m 8192 --n 8192 --k 65536 --iters 5 [--fill zeros]"""
import argparse, ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="auto")
ap.add_argument("--m", type=int, default=8192)
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--k", type=int, default=65536)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--fill", default="uniform")
a = ap.parse_args()
be = ta.get_hip_backend()
if a.fill == "zeros":
  A = be.zeros((a.m * a.k,), dtype=ta.bfloat16); B = be.zeros((a.n * a.k,), dtype=ta.bfloat16)
else:
  A = be.device_random((a.m * a.k,), dtype=ta.bfloat16, seed=1, normal=False, a=-1.0, b=1.0)
  B = be.device_random((a.n * a.k,), dtype=ta.bfloat16, seed=2, normal=False, a=-1.0, b=1.0)
C = DeviceTensor.empty((a.m, a.n), _lib.BF16)
_lib.check(be.lib.tnh_gemm_set_variant(a.variant.encode()))
def call():
  _lib.check(be.lib.tnh_gemm(_lib.BF16, _lib.BF16, 0, 1, a.m, a.n, a.k, ctypes.c_void_p(A.ptr), a.k,
                             ctypes.c_void_p(B.ptr), a.k, ctypes.c_void_p(C.ptr), a.n, 1, 0, 0, 0))
call(); be.synchronize()
s = _lib.Event().record()
for _ in range(a.iters): call()
e = _lib.Event().record(); e.synchronize()
ms = s.elapsed_ms(e) / a.iters
print(json.dumps({"variant": a.variant, "kernel": be.lib.tnh_gemm_last_kernel().decode(), "m": a.m, "n": a.n, "k": a.k,
                  "fill": a.fill, "ms": ms, "tflops": 2.0 * a.m * a.n * a.k / ms / 1e9}))
