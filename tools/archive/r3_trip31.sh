#!/bin/bash
# where do the 5 ms between the band-SVD probe and the bench's split_node go?
set -u
O=gpurun_out/${1:-r3t31}
mkdir -p $O
timeout 600 python - <<'PY' | tee $O/svd_host.txt
import time, numpy as np, tensornetwork_amd as ta, bench
from tensornetwork_amd import _lib
be = ta.get_hip_backend()
n, k = 4096, 256
mat = be.device_random((n, n), dtype=np.float32, seed=3, normal=True)
def wall(fn, reps=3):
  out = []
  for _ in range(reps):
    be.synchronize(); t0 = time.perf_counter(); r = fn(); be.synchronize(); out.append((time.perf_counter() - t0) * 1e3); del r
  return out
print("be.svd            ", ["%.2f" % x for x in wall(lambda: be.svd(mat, 1, max_singular_values=k))])
x = be.reshape(mat, (16,) * 6)
def split():
  node = ta.Node(x, backend=be)
  return ta.split_node(node, [node[i] for i in (0, 1, 2)], [node[i] for i in (3, 4, 5)], max_singular_values=k)
print("split_node natural", ["%.2f" % x for x in wall(split)])
import cProfile, pstats, io
pr = cProfile.Profile(); be.synchronize(); pr.enable(); r = split(); be.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
PY
timeout 300 python tools/svd_band_probe.py 4096 256 gauss --no-check > $O/probe.json 2>> $O/probe.err
python -c "
import json; r=json.load(open('$O/probe.json')); print('probe: factor %.2f vectors %.2f total %.2f ms'%(r['rep2']['factor_ms'],r['rep2']['vectors_ms'],r['rep2']['total_ms']))"
