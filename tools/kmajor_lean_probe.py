"""Round 6: the k-major B operand of tensordot read in place by the interleaved whole-K-tile lean loop (B_KT in
tnh_gemm_bf16.hip) against (w0) the round-5 k-major loop and (pass) ONE K1 pass + the NT kernel, on the sweep's L0
shapes and the D = 512 row.  Per shape: seconds per tensordot of each form (K1 pass included where there is one),
TFLOP/s, bit-identity of the three results.
  python tools/kmajor_lean_probe.py [--rows 96,128,192,256,512row] [--iters 6] [--fill normal|zeros]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="96,128,192,512row")
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--fill", default="normal")
ap.add_argument("--out", default="")
a = ap.parse_args()
be = ta.get_hip_backend()


def operands(row):
  if row == "512row":
    sa, sb, axes = (64, 128, 512, 512), (512, 512, 128, 64), ([2, 3], [0, 1])
  else:
    d = int(row)
    sa = sb = (d, d, d, d)
    axes = ([2, 3], [0, 1])
  if a.fill == "zeros":
    return be.zeros(sa, dtype=ta.bfloat16), be.zeros(sb, dtype=ta.bfloat16), axes
  k = sa[2] * sa[3]
  A = be.device_random(sa, dtype=ta.bfloat16, seed=11, normal=True, a=0.0, b=k ** -0.5)
  B = be.device_random(sb, dtype=ta.bfloat16, seed=12, normal=True, a=0.0, b=1.0)
  return A, B, axes


def timed(A, B, axes, iters):
  out = be.tensordot(A, B, axes)
  be.synchronize()
  before = be.permute_launches
  s = _lib.Event().record()
  for _ in range(iters):
    out = None
    out = be.tensordot(A, B, axes)
  e = _lib.Event().record()
  e.synchronize()
  return out, s.elapsed_ms(e) / iters, (be.permute_launches - before) / iters, be.lib.tnh_gemm_last_kernel().decode()


lines = []
for row in a.rows.split(","):
  A, B, axes = operands(row)
  m = A.shape[0] * A.shape[1]
  n = B.shape[2] * B.shape[3]
  k = A.shape[2] * A.shape[3]
  flop = 2.0 * m * n * k
  rec = {"row": row, "gemm": [m, n, k], "fill": a.fill}
  keep = (be.kmajor_inplace_penalty, be.inplace_max_bytes)
  keep_tw = be.kmajor_tile_walk
  results = {}
  for form in ("pass", "kt", "w0", "kt", "pass"):
    if form == "pass":
      be.kmajor_inplace_penalty, be.inplace_max_bytes = 1e9, keep[1]      # never in place
      be.kmajor_tile_walk = False
      knob = b"auto"
    else:
      be.kmajor_inplace_penalty, be.inplace_max_bytes = 0.0, 1 << 40      # always in place
      knob = b"auto" if form == "kt" else b"auto:w0"
    _lib.check(be.lib.tnh_gemm_set_variant(knob))
    try:
      out, ms, perms, kernel = timed(A, B, axes, a.iters)
    finally:
      _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))
      be.kmajor_inplace_penalty, be.inplace_max_bytes = keep
      be.kmajor_tile_walk = keep_tw
    rec.setdefault(form, []).append({"ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1), "permutes": perms, "kernel": kernel})
    if form not in results:
      # a sample of the result: rows 0, 1 and the last one (whole rows, every column tile)
      results[form] = np.stack([np.asarray(be.getitem(out.view((m, n)), (int(r),))) for r in (0, 1, m // 2 + 3, m - 1)])
    out = None
  rec["kt_equals_w0"] = bool(np.array_equal(results["kt"], results["w0"]))
  rec["kt_equals_pass"] = bool(np.array_equal(results["kt"], results["pass"]))
  lines.append(rec)
  print(json.dumps(rec), flush=True)
  A = B = None
if a.out:
  with open(a.out, "w") as f:
    for r in lines:
      f.write(json.dumps(r) + "\n")
