"""config-2 L0 at D = 128, b read in place (k-major, nn view kernel) or through one K1 pass (NT view kernel): the
workload of a rocprofv3 --pmc comparison.  argv[1] = "inplace" | "permute"."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
import bench
be = ta.get_hip_backend()
be.kmajor_inplace_penalty = 0.0 if sys.argv[1] == "inplace" else 0.10
A, B = bench.make_nodes(ta, be, 128, "L0", seed=7, fill="normal")
for _ in range(6):
  out = bench.one_step(ta, be, A, B, "L0")
  del out
be.synchronize()
print(be.lib.tnh_gemm_last_kernel().decode())
