"""cProfile of one two-site DMRG sweep (XXZ, N = 32, D = 256, f32) on the GPU: where the host time goes."""
import cProfile, pstats, sys, os, time, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import mps as tmps
be = ta.get_hip_backend()
n, D, dt = 32, (int(sys.argv[1]) if len(sys.argv) > 1 else 256), np.float32
mpo = tmps.xxz_mpo(be, np.ones(n - 1), np.ones(n - 1), np.zeros(n), dtype=dt)
state = tmps.FiniteMPS.random([2] * n, [16] * (n - 1), dt, be, seed=1)
dm = tmps.FiniteDMRG(state, mpo)
dm.run_two_site(max_bond_dim=D, num_sweeps=2, num_krylov_vecs=10)
be.synchronize()
import collections
seen = collections.Counter()
_svd = be.svd
def svd_logged(t, *a, **k):
  out = _svd(t, *a, **k)
  seen[(tuple(t.shape), be.last_svd_path, be.last_svd_band_status)] += 1
  return out
be.svd = svd_logged
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
dm.run_two_site(max_bond_dim=D, num_sweeps=1, num_krylov_vecs=10, precision=0.0)
be.synchronize()
pr.disable()
print("sweep seconds", time.perf_counter() - t0)
for key, cnt in sorted(seen.items(), key=lambda kv: -kv[1])[:12]:
  print("  svd", key, cnt)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:5000])
