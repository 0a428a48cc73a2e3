"""configs[3]: <psi|psi> of a 16-site MPS (32 nodes: kets + conjugates), d = 2, bulk bond D, contracted
with contractors.greedy -- launch-latency-bound (~1e9 flop at D = 512), so also the heavier
d = 4 variant.  GPU (hip backend, with and without hipGraph replay) beside the NumPy oracle backend.
  python tests/perf_mps_chain.py [--D 512] [--d 2,4]
Lives under tests/ (not collected by pytest): it times the CPU oracle beside the GPU path, and only
tests/, smoke() and bench.py's cpu_baseline leg may import oracle/."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root (oracle/ lives there)
import tensornetwork_amd as ta
from tensornetwork_amd import contractors, workloads as wl
from oracle import numpy_oracle as orc
ap = argparse.ArgumentParser(); ap.add_argument("--D", type=int, default=512); ap.add_argument("--d", default="2,4")
a = ap.parse_args()
hip = ta.get_hip_backend()
for d in [int(x) for x in a.d.split(",")]:
  kets = wl.mps_tensors(16, d, a.D, seed=5, dtype=np.float32)
  dev = [hip.convert_to_tensor(k) for k in kets]
  def run(be, tensors):
    return contractors.greedy(wl.mps_overlap_network(be, tensors)).tensor
  run(hip, dev); hip.synchronize()
  reps = 5
  t0 = time.perf_counter()
  for _ in range(reps): out = run(hip, dev)
  hip.synchronize(); tg = (time.perf_counter() - t0) / reps
  g = hip.capture(lambda *ts: run(hip, list(ts)), *dev)
  g.launch(); hip.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps): o2 = g.launch()
  hip.synchronize(); tr = (time.perf_counter() - t0) / reps
  be = orc.OracleBackend()
  run(be, kets)
  t0 = time.perf_counter(); ref = run(be, kets); tc = time.perf_counter() - t0
  print(json.dumps({"sites": 16, "d": d, "D": a.D, "dtype": "f32", "gpu_eager_ms": tg * 1e3, "gpu_graph_replay_ms": tr * 1e3,
                    "cpu_ms": tc * 1e3, "value_gpu": float(np.asarray(out)), "value_graph": float(np.asarray(o2[0] if isinstance(o2, (list, tuple)) else o2)),
                    "value_cpu": float(ref)}), flush=True)
  g.close()
