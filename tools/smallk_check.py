"""The K <= 16 store-stream kernel (tnh_gemm_smallk.hip; TNH_GEMM_SMALLK=1) against float64 and against the tile kernels
on a few shapes, with timings.   TNH_GEMM_SMALLK=1 python tools/smallk_check.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from oracle import numpy_oracle as orc

be = ta.get_hip_backend()
rng = np.random.default_rng(9)
ok = True
for dtype, (m, n, k) in [("bf16", (1000, 20008, 12)), ("f16", (144, 65536, 4)), ("bf16", (1728, 24576, 16)), ("bf16", (200, 40000, 8))]:
  a = rng.standard_normal((m, k)).astype(np.float32)
  b = rng.standard_normal((n, k)).astype(np.float32)
  if dtype == "bf16":
    a, b = orc.round_bf16(a), orc.round_bf16(b)
    da, db = be.to_bfloat16(a), be.to_bfloat16(b)
  else:
    a, b = a.astype(np.float16), b.astype(np.float16)
    da, db = be.convert_to_tensor(a), be.convert_to_tensor(b)
  got = np.asarray(be.tensordot(da, db, [[1], [1]])).astype(np.float64)
  kernel = be.lib.tnh_gemm_last_kernel().decode()
  ref = a.astype(np.float64) @ b.astype(np.float64).T
  ulp = 2.0**-7 if dtype == "bf16" else 2.0**-10      # relative spacing at the bottom of a binade
  err = float(np.max(np.abs(got - ref) / (np.abs(ref) * ulp + 1e-3 * ulp)))
  good = err <= 0.52 and kernel.startswith("bf16_smallk")
  ok = ok and good
  print(json.dumps({"dtype": dtype, "m": m, "n": n, "k": k, "kernel": kernel, "max_err_in_ulps": err, "ok": good}), flush=True)

# the product of the D = 12 network: 1728 x 248 832 x 12, both lowerings timed
a = be.device_random((1728, 12), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=1.0)
b = be.device_random((248832, 12), dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=1.0)
def timed(reps=10):
  out = be.tensordot(a, b, [[1], [1]]); del out
  be.synchronize(); t0 = time.perf_counter()
  for _ in range(reps):
    out = be.tensordot(a, b, [[1], [1]]); del out
  be.synchronize()
  return (time.perf_counter() - t0) / reps
t_small = timed()
name_small = be.lib.tnh_gemm_last_kernel().decode()
_lib.check(be.lib.tnh_gemm_set_variant(b"bf16_ragged"))
try:
  t_tile = timed()
  name_tile = be.lib.tnh_gemm_last_kernel().decode()
finally:
  _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
print(json.dumps({"product": "1728 x 248832 x 12 bf16", "smallk_us": t_small * 1e6, "smallk_kernel": name_small, "tile_us": t_tile * 1e6,
                  "tile_kernel": name_tile, "TBps_smallk": 2 * 1728 * 248832 / t_small / 1e12, "all_ok": ok}), flush=True)
