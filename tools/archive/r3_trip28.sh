#!/bin/bash
# band SVD / QR with the V rows formed inside the W / Y passes: tests, A/B against the separate formv launches
set -u
O=gpurun_out/${1:-r3t28}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log
for f in 0 1 0 1; do
  TNH_SVDB_FORMV=$f timeout 300 python tools/svd_band_probe.py 4096 256 gauss > $O/probe_f$f.json 2>> $O/probe.err
  python -c "
import json; r=json.load(open('$O/probe_f$f.json')); print('formv_fused=$f: factor %.2f vectors %.2f total %.2f ms'%(r['rep2']['factor_ms'],r['rep2']['vectors_ms'],r['rep2']['total_ms']), r['rep2']['status'], 's_err %.2e orth %.2e'%(r['s_err_over_s0'], r['orth_u']))"
done
timeout 300 python tools/qr_sizes_probe.py 2>&1 | tail -6
