"""Where the host time of a small contraction goes: cProfile over contract_between at D = 32 (GEMM 1024^3, ~5 us of
kernel time) and over an MPS <psi|psi> chain.  python tools/host_overhead_probe.py"""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta

be = ta.get_hip_backend()
D = 32
A = be.device_random((D,) * 4, dtype=ta.bfloat16, seed=1, normal=True)
B = be.device_random((D,) * 4, dtype=ta.bfloat16, seed=2, normal=True)


def step():
  a, b = ta.Node(A, backend=be), ta.Node(B, backend=be)
  a[2] ^ b[0]
  a[3] ^ b[1]
  return ta.contract_between(a, b)


def raw():
  return be.tensordot(A, B, [[2, 3], [0, 1]])


for name, fn in (("contract_between", step), ("backend.tensordot", raw)):
  for _ in range(50): fn()
  be.synchronize()
  t0 = time.perf_counter()
  for _ in range(2000): fn()
  be.synchronize()
  print(name, "us per call:", round((time.perf_counter() - t0) / 2000 * 1e6, 1))
pr = cProfile.Profile()
pr.enable()
for _ in range(2000): step()
be.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])

# ---- round 6: the configs[3] chain (16-site MPS overlap, D = 512, d = 2, f32) under the same lens
from tensornetwork_amd import contractors, workloads as wl
kets = wl.mps_tensors(16, 2, 512, seed=5, dtype=np.float32)
dev = [be.convert_to_tensor(k) for k in kets]
def chain():
  return contractors.greedy(wl.mps_overlap_network(be, dev)).tensor
for _ in range(5): chain()
be.synchronize()
t0 = time.perf_counter()
for _ in range(50): chain()
be.synchronize()
print("mps chain eager ms per call:", round((time.perf_counter() - t0) / 50 * 1e3, 3))
t0 = time.perf_counter()
for _ in range(50): wl.mps_overlap_network(be, dev)
print("  of which building the 32-node network:", round((time.perf_counter() - t0) / 50 * 1e3, 3), "ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(50): chain()
be.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
