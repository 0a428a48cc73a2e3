"""Two-site DMRG bond-dimension sweep: XXZ chain, GPU backend vs the NumPy oracle backend on the host.
  python tests/perf_dmrg.py [--n 32] [--bonds 64,128,256] [--dtype float32] [--cpu-max 128]
Lives under tests/ (not collected by pytest): it times the CPU oracle beside the GPU path, and only
tests/, smoke() and bench.py's cpu_baseline leg may import oracle/."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root (oracle/ lives there)
import tensornetwork_amd as ta
from tensornetwork_amd import mps as tmps
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--bonds", default="64,128,256")
ap.add_argument("--dtype", default="float32")
ap.add_argument("--cpu-max", type=int, default=128)
ap.add_argument("--deferred", action="store_true", help="device-resident Lanczos coefficients (krylov.eigsh_lanczos_deferred)")
a = ap.parse_args()
dt = np.dtype(a.dtype).type
n = a.n

def run(be, D, sweeps):
  mpo = tmps.xxz_mpo(be, np.ones(n - 1), np.ones(n - 1), np.zeros(n), dtype=dt)
  state = tmps.FiniteMPS.random([2] * n, [min(D, 16)] * (n - 1), dt, be, seed=1)
  dm = tmps.FiniteDMRG(state, mpo)
  dm.deferred_lanczos = a.deferred
  dm.run_two_site(max_bond_dim=D, num_sweeps=2, num_krylov_vecs=10)   # grow the bonds to D
  if hasattr(be, "synchronize"): be.synchronize()
  t0 = time.perf_counter()
  e = dm.run_two_site(max_bond_dim=D, num_sweeps=sweeps, num_krylov_vecs=10, precision=0.0)
  if hasattr(be, "synchronize"): be.synchronize()
  return (time.perf_counter() - t0) / sweeps, float(np.real(e)), max(state.bond_dimensions)

hip = ta.get_hip_backend()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import numpy_oracle as orc
for D in [int(x) for x in a.bonds.split(",")]:
  t, e, dmax = run(hip, D, 1)
  rec = {"n": n, "D": D, "dtype": a.dtype, "gpu_s_per_sweep": t, "energy": e, "max_bond": dmax}
  if D <= a.cpu_max:
    tc, ec, _ = run(orc.OracleBackend(), D, 1)
    rec.update({"cpu_s_per_sweep": tc, "cpu_energy": ec})
  print(json.dumps(rec), flush=True)
