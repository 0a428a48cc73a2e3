"""Gather GEMM against permute + streaming GEMM on the operand shapes of the D = 12 network (bench.py's
`gather_gemm` leg alone): a 144 x 144 tensor takes two bonds off a rank-8 intermediate (12^8 = 430 M elements) whose
contracted axes sit at different places.
  python tools/gather_probe.py [--reps 5] [--no-verify]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--no-verify", action="store_true")
a = ap.parse_args()
be = ta.get_hip_backend()
cases = dict(bench.GATHER_CASES)
cases["k67_trailing_never_gathered"] = [6, 7]
out = bench.gather_gemm_bench(ta, be, not a.no_verify, reps=a.reps, cases=cases)
for row in out.pop("rows"):
  print(json.dumps(row), flush=True)
print(json.dumps(out), flush=True)
