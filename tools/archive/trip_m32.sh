#!/bin/bash
set -u
export TMPDIR=/tmp
# correctness of the M32 variant first (speed-path test parametrised with p3), then A/B timings
timeout 300 python - <<'PY'
import numpy as np, sys, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
be = ta.get_hip_backend()
rng = np.random.default_rng(0)
for (m, n, k) in [(512, 384, 256), (1024, 768, 512), (2048, 2304, 1088)]:
  a = ta.round_to_bf16(rng.standard_normal((m, k))); b = ta.round_to_bf16(rng.standard_normal((n, k)))
  _lib.check(be.lib.tnh_gemm_set_variant(b"bf16_256pp:p3"))
  out = np.asarray(be.tensordot(be.to_bfloat16(a), be.to_bfloat16(b), [[1], [1]]))
  _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
  err = np.abs(out - a @ b.T).max() / np.sqrt(k)
  print("m32 check", m, n, k, "max err/sqrt(k)", err, "OK" if err < 2e-2 else "FAIL")
PY
for v in "bf16_256pp:p2" "bf16_256pp:p3" "bf16_256pp:p2" "bf16_256pp:p3"; do
  timeout 120 python tools/gemm_one.py --variant $v --m 8192 --n 8192 --k 65536 --iters 5 --fill uniform 2>&1 | tail -1
done
for v in "bf16_256pp:p2" "bf16_256pp:p3"; do
  timeout 120 python tools/gemm_one.py --variant $v --m 8192 --n 8192 --k 65536 --iters 5 --fill zeros 2>&1 | tail -1
done
