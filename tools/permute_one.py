"""Time one permute: python tools/permute_one.py --shape 16,16,16 --perm 2,1,0 [--dtype bf16]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--shape"); ap.add_argument("--perm"); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
be = ta.get_hip_backend()
shape = tuple(int(v) for v in a.shape.split(",")); perm = tuple(int(v) for v in a.perm.split(","))
dt = {"bf16": ta.bfloat16, "f32": np.float32, "f64": np.float64}[a.dtype]
x = be.device_random(shape, dtype=dt, seed=1, normal=True)
y = be.transpose(x, perm); be.synchronize()
s = _lib.Event().record()
for _ in range(a.iters): y = be.transpose(x, perm)
e = _lib.Event().record(); e.synchronize()
ms = s.elapsed_ms(e) / a.iters
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("TNH_")}, "ms": ms, "TBps": 2 * x.size * x.itemsize / ms / 1e9}))
