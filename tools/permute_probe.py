"""K1 permute bandwidth on the GPU box (algorithmic bytes = 2 * numel * itemsize).
  python tools/permute_probe.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
be = ta.get_hip_backend()
cases = [("bf16", ta.bfloat16, (128,) * 4, (2, 3, 0, 1)), ("bf16", ta.bfloat16, (128,) * 4, (1, 3, 0, 2)),
         ("bf16", ta.bfloat16, (16384, 16384), (1, 0)), ("f32", np.float32, (16384, 16384), (1, 0)),
         ("f32", np.float32, (128,) * 4, (2, 3, 0, 1)), ("f32", np.float32, (16,) * 6, (0, 2, 4, 1, 3, 5)),
         ("f64", np.float64, (8192, 8192), (1, 0))]
for name, dt, shape, perm in cases:
  x = be.device_random(shape, dtype=dt, seed=1, normal=True)
  y = be.transpose(x, perm); be.synchronize()
  s = _lib.Event().record()
  for _ in range(10):
    y = be.transpose(x, perm)
  e = _lib.Event().record(); e.synchronize()
  ms = s.elapsed_ms(e) / 10
  nbytes = 2 * x.size * x.itemsize
  print(json.dumps({"dtype": name, "shape": shape, "perm": perm, "ms": ms, "TBps": nbytes / ms / 1e9}), flush=True)
