"""Round 6: what a k-major B ([K][pitch], rows of N contiguous) costs the view GEMM beyond its instructions -- the
same product with B's row pitch padded (power-of-two pitches put every k row of a column tile on the same channels),
under both tile rasters, at several K; against the NT form (B as [N][K]) of the same shape.
  python tools/kmajor_mem_probe.py --m 8192 --n 8192 --k 262144 --pads 0,64,128,1024 [--fill zeros] [--rasters 0,1]"""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
from tensornetwork_amd import _lib
from tensornetwork_amd.device_tensor import DeviceTensor

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=8192)
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--k", type=int, default=262144)
ap.add_argument("--pads", default="0,64,128,1024")
ap.add_argument("--rasters", default="-1")
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--fill", default="normal")
ap.add_argument("--wrap-b", type=int, default=0, help="k-major B: the contraction index wraps around after this many rows "
                "(the loop keeps its length, B's footprint shrinks)")
ap.add_argument("--freeze", default="", help="a / b / ab: that operand re-reads its first K-tile for the whole contraction "
                "(contraction runs of 64 with stride 0 between them): its requests always hit, the L2 misses left are the other operand's")
a = ap.parse_args()
be = ta.get_hip_backend()
M, N, K = a.m, a.n, a.k
flop = 2.0 * M * N * K


def fill(shape, seed, sigma):
  if a.fill == "zeros":
    return be.zeros(shape, dtype=ta.bfloat16)
  return be.device_random(shape, dtype=ta.bfloat16, seed=seed, normal=True, a=0.0, b=sigma)


A = fill((M, K), 1, K ** -0.5)
C = DeviceTensor.empty((M, N), _lib.BF16)
va = _lib.OperandView(M, K, 0, 64, 1, 0) if "a" in a.freeze else _lib.OperandView(M, K, 0, K, 1, 0)


def run(Bt, vb, knob):
  _lib.check(be.lib.tnh_gemm_set_variant(knob.encode()))
  try:
    def call():
      _lib.check(be.lib.tnh_gemm_view(_lib.BF16, _lib.BF16, M, N, K, ctypes.c_void_p(A.ptr), ctypes.byref(va),
                                      ctypes.c_void_p(Bt.ptr), ctypes.byref(vb), ctypes.c_void_p(C.ptr), N), "view")
    call()
    be.synchronize()
    s = _lib.Event().record()
    for _ in range(a.iters):
      call()
    e = _lib.Event().record()
    e.synchronize()
    return s.elapsed_ms(e) / a.iters, be.lib.tnh_gemm_last_kernel().decode()
  finally:
    _lib.check(be.lib.tnh_gemm_set_variant(b"auto"))


rasters = [int(r) for r in a.rasters.split(",")]
Bn = fill((N, K), 2, 1.0)
for r in rasters:
  knob = "auto" if r < 0 else f"auto:r{r}"
  ms, kern = run(Bn, _lib.OperandView(N, K, 0, 64, 1, 0) if "b" in a.freeze else _lib.OperandView(N, K, 0, K, 1, 0), knob)
  print(json.dumps({"form": "nt", "freeze": a.freeze, "gemm": [M, N, K], "raster": r, "fill": a.fill, "ms": round(ms, 4),
                    "tflops": round(flop / ms / 1e9, 1), "kernel": kern}), flush=True)
Bn = None
for pad in [int(p) for p in a.pads.split(",")]:
  pitch = N + pad
  Bk = fill((K, pitch), 3, 1.0)
  vb = _lib.OperandView(N, 1, 0, 64, pitch, 0) if "b" in a.freeze else _lib.OperandView(N, 1, 0, K, pitch, 0)
  if a.wrap_b:
    vb = _lib.OperandView(N, 1, 0, a.wrap_b, pitch, 0)
  for r in rasters:
    for knob0 in ("auto", "auto:w0"):
      knob = knob0 if r < 0 else f"{knob0}:r{r}"
      ms, kern = run(Bk, vb, knob)
      print(json.dumps({"form": "kn", "freeze": a.freeze, "wrap_b": a.wrap_b, "loop": {"auto": "kt", "auto:w0": "w0"}.get(knob0, knob0), "gemm": [M, N, K], "pitch": pitch,
                        "raster": r, "fill": a.fill, "ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1), "kernel": kern}),
            flush=True)
  Bk = None
