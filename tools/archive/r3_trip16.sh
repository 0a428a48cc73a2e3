#!/bin/bash
# cluster Gram-Schmidt, complex via embedding, full default bench (new verified legs)
set -u
O=gpurun_out/${1:-r3t16}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -q --timeout 600 > $O/pytest_a.log 2>&1; echo "pytest band+linalg rc=$?" | tee $O/trip.log
tail -15 $O/pytest_a.log
python - <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import tensornetwork_amd as ta
be = ta.get_hip_backend()
rng=np.random.default_rng(1)
a=(rng.standard_normal((1024,1024))+1j*rng.standard_normal((1024,1024))).astype(np.complex64)
d=be.convert_to_tensor(a)
for flag in (True, False):
  be.svd_band = flag
  be.svd(d,1,max_singular_values=64); be.synchronize()
  t0=time.perf_counter(); be.svd(d,1,max_singular_values=64); be.synchronize()
  print("complex64 1024^2 keep 64: %.2f ms"%((time.perf_counter()-t0)*1e3), be.last_svd_path, flush=True)
be.svd_band = True
x=be.convert_to_tensor(rng.standard_normal((1024,1024)).astype(np.float32))
be.svd(x,1,max_singular_values=64); be.synchronize()
t0=time.perf_counter(); be.svd(x,1,max_singular_values=64); be.synchronize()
print("f32 1024^2 keep 64: %.2f ms"%((time.perf_counter()-t0)*1e3), be.last_svd_path)
PY
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/trip.log
tail -3 $O/bench.err
python - <<PY
import json
r=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","n_gpus")}, r["roofline"]["frac"])
v=r.get("verified",{})
for k,x in v.items():
  if isinstance(x,dict):
    print(k, {kk:vv for kk,vv in x.items() if kk in ("ok","rms_rel_err","model_rms","err_over_tol","n_values","rel_err_of_the_sum","rel_diff_of_the_scalars","worst_s_err_over_s0","error")})
  else: print(k,x)
print("svd", {k:r["svd"].get(k) for k in ("seconds","gbps","path","error")})
for row in r.get("bond_sweep",[]): print(row.get("D"),row.get("layout"),round(row.get("tflops",0)),row.get("permute_launches"))
PY
