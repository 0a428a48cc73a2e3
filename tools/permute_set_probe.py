"""Times the large permutes one slice of the 64-node D=12 network launches (shape_trace.py's list).
  python tools/permute_set_probe.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
CASES = [
    ((12, 12, 12, 12, 12, 1, 12, 12, 12), (0, 4, 5, 6, 1, 2, 7, 8, 3)),
    ((12, 12, 12, 12, 12, 12, 1, 12, 12), (0, 2, 4, 1, 5, 6, 7, 3, 8)),
    ((12, 12, 12, 12, 12, 12, 12, 1, 12), (0, 3, 7, 8, 4, 2, 5, 6, 1)),
    ((12, 12, 12, 12, 1, 12, 12, 12, 12), (2, 3, 4, 5, 0, 6, 1, 7, 8)),
    ((12, 12, 12, 12, 12, 12, 12, 1, 12), (0, 2, 3, 4, 1, 5, 6, 7, 8)),
    ((12,) * 7, (2, 1, 3, 4, 5, 6, 0)),
    ((12,) * 7, (2, 3, 4, 0, 5, 6, 1)),
    ((12, 1, 12, 12, 12, 12, 12, 12), (1, 5, 2, 7, 3, 6, 0, 4)),
    ((12,) * 7, (2, 0, 3, 4, 1, 5, 6)),
    ((12, 12, 1, 12, 12, 12, 12, 12), (1, 2, 3, 4, 5, 6, 0, 7)),
    ((12, 12, 12, 12, 1, 12, 12, 12), (2, 3, 4, 5, 0, 6, 1, 7)),
    ((12, 12, 12, 12, 12, 12, 1, 12), (2, 0, 4, 7, 5, 3, 6, 1)),
    ((12,) * 7, (0, 1, 2, 4, 6, 5, 3)),
    ((16,) * 7, (2, 1, 3, 4, 5, 6, 0)),
    ((16,) * 7, (0, 1, 2, 4, 6, 5, 3)),
    ((8,) * 9, (0, 4, 5, 6, 1, 2, 7, 8, 3)),
]
be = ta.get_hip_backend()
for shape, perm in CASES:
  x = be.device_random(shape, dtype=ta.bfloat16, seed=1, normal=True)
  y = be.transpose(x, perm); y = be.transpose(x, perm); be.synchronize()   # both output blocks of the loop come from the pool
  s = _lib.Event().record()
  for _ in range(5): y = be.transpose(x, perm)
  e = _lib.Event().record(); e.synchronize()
  ms = s.elapsed_ms(e) / 5
  print(json.dumps({"shape": shape, "perm": perm, "elems": x.size, "ms": round(ms, 4), "TBps": round(2 * x.size * 2 / ms / 1e9, 3)}), flush=True)
  del x, y
