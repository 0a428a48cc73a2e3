"""configs[3] under the microscope (round 6): every GEMM / permute the 16-site MPS overlap (D = 512, d = 2, f32)
launches through contractors.greedy, with the kernel the library picked and its time under events.
  python tools/mps_chain_shapes.py [--D 512] [--d 2]"""
import argparse, ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import contractors, workloads as wl, _lib

ap = argparse.ArgumentParser(); ap.add_argument("--D", type=int, default=512); ap.add_argument("--d", type=int, default=2)
a = ap.parse_args()
be = ta.get_hip_backend()
real = be.lib
LOG = []


class Proxy:
  def __getattr__(self, name):
    fn = getattr(real, name)
    if name in ("tnh_gemm", "tnh_gemm_ex", "tnh_gemm_view", "tnh_permute", "tnh_sum_mid", "tnh_trace_last2"):
      def wrapped(*args):
        real.tnh_sync()
        t0 = time.perf_counter()
        rc = fn(*args)
        real.tnh_sync()
        dt = time.perf_counter() - t0
        if name == "tnh_gemm":
          LOG.append((name, [int(x) for x in args[2:7]], real.tnh_gemm_last_kernel().decode(), dt * 1e6))
        elif name == "tnh_permute":
          LOG.append((name, [int(args[2])], "", dt * 1e6))
        else:
          LOG.append((name, [], real.tnh_gemm_last_kernel().decode() if "gemm" in name else "", dt * 1e6))
        return rc
      return wrapped
    return fn


kets = wl.mps_tensors(16, a.d, a.D, seed=5, dtype=np.float32)
dev = [be.convert_to_tensor(k) for k in kets]
def run():
  return contractors.greedy(wl.mps_overlap_network(be, dev)).tensor
run(); be.synchronize()
be._lib = Proxy()
run()
be._lib = real
for rec in LOG:
  print(json.dumps({"call": rec[0], "ta,tb,m,n,k": rec[1], "kernel": rec[2], "us_with_sync": round(rec[3], 1)}))
print(json.dumps({"calls": len(LOG), "sum_us": round(sum(r[3] for r in LOG), 1)}))
