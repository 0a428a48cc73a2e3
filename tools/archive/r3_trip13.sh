#!/bin/bash
# pJ / flop table of the bf16 GEMM variants under the power cap; band SVD at 512 / 768; helper baselines
set -u
O=gpurun_out/${1:-r3t13}
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python tools/power_probe.py --seconds 1.5 --shapes 8192x8192x8192,8192x8192x65536 --fills normal,zeros \
   --variants auto,bf16_256pp:p4,bf16_256pp:p3,bf16_256pp:p6,bf16_256pp:r0,bf16_256pp:r2 > $O/power_variants.jsonl 2> $O/power.err; echo "power rc=$?"
python - <<PY
import json
print("%-22s %-18s %-7s %8s %7s %7s %9s"%("shape","variant","fill","TF","W","MHz","pJ/flop"))
for l in open("$O/power_variants.jsonl"):
  r=json.loads(l)
  if "tflops" not in r: continue
  print("%-22s %-18s %-7s %8.0f %7.0f %7.0f %9.3f"%("%dx%dx%d"%(r["m"],r["n"],r["k"]),r["variant"],r["fill"],r["tflops"],r.get("power_mean_w") or 0,r.get("sclk_mean_mhz") or 0,r.get("pj_per_flop") or 0))
PY
tail -3 $O/power.err
for args in "512 32 gauss" "512 32 graded" "768 48 gauss" "1024 64 gauss"; do
  TNH_SVD_BAND_MIN=256 timeout 120 python tools/svd_band_probe.py $args >> $O/probe_small.jsonl 2>> $O/probe.err
done
python - <<PY
import json
for l in open("$O/probe_small.jsonl"):
  r=json.loads(l); print(r["n"],r["kind"],"band direct total %.2f ms"%r["rep2"]["total_ms"],r["rep2"]["status"],"s_err %.1e orth %.1e"%(r["s_err_over_s0"],r["orth_u"]))
PY
python - <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import tensornetwork_amd as ta
be = ta.get_hip_backend()
be.svd_band_min = 10**9
for n,k,kind in ((512,32,"gauss"),(512,32,"graded"),(768,48,"gauss")):
  rng=np.random.default_rng(1)
  if kind=="gauss": a=rng.standard_normal((n,n)).astype(np.float32)
  else:
    qu,_=np.linalg.qr(rng.standard_normal((n,n))); qv,_=np.linalg.qr(rng.standard_normal((n,n)))
    a=((qu*2.0**(-np.arange(n)/32.0))@qv.T).astype(np.float32)
  d=be.convert_to_tensor(a); be.svd(d,1,max_singular_values=k); be.synchronize()
  t0=time.perf_counter(); be.svd(d,1,max_singular_values=k); be.synchronize()
  print(n,kind,"jacobi %.2f ms"%((time.perf_counter()-t0)*1e3), be.last_svd_path, be.last_svd_sweeps)
PY
