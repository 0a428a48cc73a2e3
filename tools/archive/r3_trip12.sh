#!/bin/bash
# QR sign fix regression + full default bench (new verified legs)
set -u
O=gpurun_out/${1:-r3t12}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_linalg.py tests/test_gpu_svd_band.py -q --timeout 600 > $O/pytest_a.log 2>&1; echo "pytest linalg+band rc=$?" | tee $O/trip.log
tail -6 $O/pytest_a.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/trip.log
tail -5 $O/bench.err
python - <<PY
import json
r=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","n_gpus")}, r["roofline"]["frac"])
v=r.get("verified",{})
for k,x in v.items():
  if isinstance(x,dict):
    print(k, {kk:vv for kk,vv in x.items() if kk in ("ok","err_over_bound","rel_err","path_depth","worst_s_err_over_s0","bound","abs_err","error")})
  else: print(k,x)
print("mera", {k:r["mera"].get(k) for k in ("seconds","tflops","permute_launches","error")} if "mera" in r else None)
print("mera64", {k:r["mera_chi64"].get(k) for k in ("tflops_1gpu","error")} if "mera_chi64" in r else None)
print("svd", {k:r["svd"].get(k) for k in ("seconds","gbps","path","error")})
for row in r.get("bond_sweep",[]): print(row.get("D"),row.get("layout"),round(row.get("tflops",0)),row.get("permute_launches"))
PY
