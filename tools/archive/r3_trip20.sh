#!/bin/bash
set -u
python - <<'PY'
import json, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
be = ta.get_hip_backend()
def rate(x, perm, reps=10):
  be.transpose(x, perm); be.synchronize()
  s=_lib.Event().record()
  for _ in range(reps): y=be.transpose(x, perm); del y
  e=_lib.Event().record(); e.synchronize()
  return 2*x.nbytes/(s.elapsed_ms(e)/reps)/1e6
xf = be.device_random((16384, 16384), dtype=np.float32, seed=1)
x6 = be.device_random((16,)*6, dtype=np.float32, seed=1)
xd = be.device_random((8192, 8192), dtype=np.float64, seed=1)
for order in ("1", "4", "8", "16", "32"):
  os.environ["TNH_PERMUTE_ORDER_T"] = order
  print("order", order, "f32 16384^2 T %6.0f  f32 (16,)^6 mixed %6.0f  f64 8192^2 T %6.0f" % (rate(xf,(1,0)), rate(x6,(0,2,4,1,3,5)), rate(xd,(1,0))), flush=True)
os.environ.pop("TNH_PERMUTE_ORDER_T")
import bench
for r in bench.helpers_bench(ta, be)[:3]: print("%-76s %7.0f GB/s"%(r["op"][:76], r["gbps"]))
PY
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "permute or transpose" 2>&1 | tail -2
