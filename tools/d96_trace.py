"""Kernel timeline of one contract_between at D = 96 (layout L0 / L1): python tools/d96_trace.py [L0|L1]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
import bench
be = ta.get_hip_backend()
layout = sys.argv[1] if len(sys.argv) > 1 else "L0"
A, B = bench.make_nodes(ta, be, 96, "L0", seed=7, fill="normal")
for _ in range(3): bench.one_step(ta, be, A, B, layout)
be.synchronize()
t0 = time.perf_counter()
for _ in range(10): bench.one_step(ta, be, A, B, layout)
be.synchronize()
print(layout, "ms per step", (time.perf_counter() - t0) / 10 * 1e3, be.lib.tnh_gemm_last_kernel().decode())
