"""config-2 layout L1 at D = 192 / 256: the strided operand `a` (two-level rows AND contraction, 2.7 / 8.6 GB) read in
place by the lean tile-granular walk against its K1 copy (HipBackend.inplace_strided_big).  GPU box only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
import bench
be = ta.get_hip_backend()
for D in (192, 256):
  A, B = bench.make_nodes(ta, be, D, "L0", seed=7, fill="normal")
  for rep in range(2):
    for flag in (True, False):
      be.inplace_strided_big = flag
      t, permutes = bench.timed_steps(be, lambda: bench.one_step(ta, be, A, B, "L1"), 3, batches=1)
      print(json.dumps({"D": D, "layout": "L1", "inplace_strided_big": flag, "rep": rep, "ms": t * 1e3,
                        "tflops": 2.0 * D**6 / t / 1e12, "permute_launches": permutes,
                        "kernel": be.lib.tnh_gemm_last_kernel().decode()}), flush=True)
  del A, B
be.inplace_strided_big = True
