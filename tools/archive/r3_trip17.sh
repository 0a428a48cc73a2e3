#!/bin/bash
# helper fast paths (elementwise f32 vectors, reduction) + complex SVD timing
set -u
O=gpurun_out/${1:-r3t17}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --timeout 600 -k "elementwise or reduc or sum or broadcast or init or cast or trace or norm" > $O/pytest_a.log 2>&1; echo "pytest rc=$?" | tee $O/trip.log
tail -5 $O/pytest_a.log
python - <<'PY'
import json, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import tensornetwork_amd as ta
import bench
be = ta.get_hip_backend()
for r in bench.helpers_bench(ta, be): print(json.dumps(r))
rng=np.random.default_rng(1)
a=(rng.standard_normal((1024,1024))+1j*rng.standard_normal((1024,1024))).astype(np.complex64)
d=be.convert_to_tensor(a)
be.svd(d,1,max_singular_values=64); be.synchronize()
t0=time.perf_counter(); be.svd(d,1,max_singular_values=64); be.synchronize()
print("complex64 1024^2 keep 64: %.2f ms"%((time.perf_counter()-t0)*1e3), be.last_svd_path, flush=True)
PY
