"""Where does split_node spend its time in the mixed edge order at n = 1024?  (bench svd sweep: 64 ms vs 9.5 natural)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta

be = ta.get_hip_backend()
for d in ((8, 8, 8), (8, 8, 16), (8, 16, 16)):
  n = d[0] * d[1] * d[2]
  mat = be.device_random((n, n), dtype=np.float32, seed=1, normal=True)
  x = be.transpose(be.reshape(mat, d + d), (0, 3, 1, 4, 2, 5))
  be.synchronize()
  def timeit(fn, reps=3):
    best = 1e9
    for _ in range(reps):
      be.synchronize(); t0 = time.perf_counter(); r = fn(); be.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
  t_perm = timeit(lambda: be.transpose(x, (0, 2, 4, 1, 3, 5)))
  t_svd = timeit(lambda: be.svd(mat, 1, max_singular_values=n // 16))
  def split(order):
    xx = x if order else be.reshape(mat, d + d)
    node = ta.Node(xx, backend=be)
    la, ra = ([0, 2, 4], [1, 3, 5]) if order else ([0, 1, 2], [3, 4, 5])
    return ta.split_node(node, [node[i] for i in la], [node[i] for i in ra], max_singular_values=n // 16)
  t_nat = timeit(lambda: split(0))
  t_mix = timeit(lambda: split(1))
  print(n, "permute %.3f ms  svd %.3f ms  split natural %.3f  mixed %.3f  path %s" % (t_perm, t_svd, t_nat, t_mix, be.last_svd_path), flush=True)
