#!/bin/bash
# permute NA=2, Sturm grid without flags, helpers, band tests
set -u
O=gpurun_out/${1:-r3t18}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_svd_band.py -q --timeout 600 -k "permute or transpose or band or tensordot or svd" > $O/pytest_a.log 2>&1; echo "pytest rc=$?" | tee $O/trip.log
tail -5 $O/pytest_a.log
python - <<'PY'
import json, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import tensornetwork_amd as ta
import bench
be = ta.get_hip_backend()
for r in bench.helpers_bench(ta, be): print("%-80s %7.0f GB/s"%(r["op"][:80], r["gbps"]))
os.environ["TNH_PERMUTE_NA1"]="1"
for r in bench.helpers_bench(ta, be)[:3]: print("NA1 %-76s %7.0f GB/s"%(r["op"][:76], r["gbps"]))
PY
timeout 300 python tools/svd_band_probe.py 4096 256 gauss > $O/probe.json 2>> $O/err.txt
python -c "
import json; r=json.load(open('$O/probe.json')); print('svd 4096: factor %.1f vectors %.1f total %.1f'%(r['rep2']['factor_ms'],r['rep2']['vectors_ms'],r['rep2']['total_ms']), r['rep2']['status'], 's_err %.2e orth %.2e'%(r['s_err_over_s0'], r['orth_u']))"
