#!/bin/bash
# D = 96 row: the bench's own timing loop vs a raw loop on the same operands
set -u
O=gpurun_out/${1:-r3t29}
mkdir -p $O
timeout 600 python - <<'PY' | tee $O/d96_bench.txt
import time, numpy as np, tensornetwork_amd as ta, bench
from tensornetwork_amd import _lib
be = ta.get_hip_backend(); lib = be.lib
ta.configure_gc(freeze=True)
for D in (96, 64, 128):
  A, B = bench.make_nodes(ta, be, D, "L0", seed=7, fill="normal")
  for rep in range(2):
    for layout in ("L0", "L1"):
      t, perm = bench.timed_steps(be, lambda: bench.one_step(ta, be, A, B, layout), 10, batches=3)
      print("D=%d %s bench loop  %.3f ms %.0f TF permutes %.1f %s" % (D, layout, t*1e3, 2.0*D**6/t/1e12, perm, lib.tnh_gemm_last_kernel().decode()), flush=True)
    def raw():
      s = _lib.Event().record()
      for _ in range(10):
        out = be.tensordot(A, B, [[2, 3], [0, 1]]); del out
      e = _lib.Event().record(); e.synchronize()
      return s.elapsed_ms(e) / 10
    raw(); ms = min(raw(), raw(), raw())
    print("D=%d L0 raw tensordot %.3f ms %.0f TF" % (D, ms, 2.0*D**6/ms/1e9), flush=True)
    t0 = time.perf_counter()
    for _ in range(200):
      a, b = ta.Node(A, backend=be), ta.Node(B, backend=be)
      a[2] ^ b[0]; a[3] ^ b[1]
    print("   node + edge creation: %.1f us per step" % ((time.perf_counter() - t0) / 200 * 1e6), flush=True)
  del A, B
PY
