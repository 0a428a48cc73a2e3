"""One pass over the helper kernels and the view GEMMs (for rocprofv3 PMC passes: LDS bank conflicts, traffic).
  python tools/helper_one.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
be = ta.get_hip_backend()
x = be.device_random((128,) * 4, dtype=ta.bfloat16, seed=42)
for _ in range(3):
  be.transpose(x, (0, 2, 1, 3))          # permute_tiled16_kernel
  be.transpose(x, (2, 3, 0, 1))
a = be.device_random((64,) * 4, dtype=ta.bfloat16, seed=1, b=1.0 / 64)
b = be.device_random((64,) * 4, dtype=ta.bfloat16, seed=2, b=1.0 / 64)
for _ in range(3):
  be.tensordot(a, b, [[2, 3], [0, 1]])   # view kernel, b k-major (ds_read_b64_tr_b16)
  be.tensordot(a, b, [[1, 3], [2, 0]])   # view kernel, two-level strides on both sides
  be.tensordot(a, b, [[0, 1], [0, 1]])   # both k-major
y = be.device_random((4096, 4096, 16), dtype=np.float32, seed=43)
for _ in range(3):
  be.sum(y, axis=1)
  be.sum(y)
be.synchronize()
print("ok", be.lib.tnh_gemm_last_kernel().decode())
