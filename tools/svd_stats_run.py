"""A few band SVD calls for rocprofv3 --kernel-trace --stats (f32 and f64, 4096^2 keep 256)."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
be = ta.get_hip_backend()
which = sys.argv[1] if len(sys.argv) > 1 else "f32"
a = be.device_random((4096, 4096), dtype=np.float32, seed=7)
if which == "f64":
  a = be.cast(a, np.float64)
for _ in range(3):
  out = be.svd(a, 1, max_singular_values=256)
be.synchronize()
print(which, be.last_svd_path, be.last_svd_band_status)
