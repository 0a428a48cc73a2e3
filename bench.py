"""Headline benchmark: one JSON line per run (contract described in the task brief).

Workload at N=1 (BASELINE.json configs[1]): ``contract_between`` of two rank-4
bf16 nodes with two shared bonds of dimension D=256 (layout L0: a[2]^b[0],
a[3]^b[1]) -> one 65536^3 GEMM on the MFMA path, operands generated in HBM.
A "step" is one such contract_between through the product path
(Node bookkeeping -> HipBackend.tensordot -> K1 permute of b -> K2 MFMA GEMM).
``value`` = 2*M*N*K*steps / wall time, TFLOP/s, inputs resident in HBM.

N>1 (one process per GPU, torch.distributed/RCCL for the barrier and the
max-over-ranks reduction only): the pairwise contraction has no exchange step,
so every rank contracts its own pair of nodes (weak scaling, no data-path
collective) and ``value`` is the aggregate.

Extra objects on the same line:
  roofline       -- the dominant kernel (bf16 MFMA GEMM) timed with HIP events on
                    the library's stream inside the timed region, vs 2.5 PFLOP/s;
                    `traffic` = HBM bytes per launch from the committed rocprofv3 PMC
                    pass of this same command (profiles/*_traffic.json), else null.
  cpu_baseline   -- the NumPy oracle (port of the reference's tensordot) timed on
                    this box's host cores on a bounded sample (same layout, smaller D).
  svd            -- split_node truncated SVD (configs[2]: (16,)*6 node -> 4096 x 4096,
                    keep 256) in the metric's GB/s, with the oracle's LAPACK SVD timed
                    on a bounded sample beside it.
  bond_sweep     -- the metric's bond-dimension sweep: contract_between of two rank-4 bf16
                    nodes at D = 32 .. 256 in the favourable (L0) and the permute-needing (L1) layout,
                    plus the north-star "D = 512" row A(64,128,512,512) . B(512,512,128,64)
                    (GEMM 8192 x 8192 x 262144), whole-path TFLOP/s each.
  dtype_sweep    -- the same contraction (backend.tensordot) at D = 64 (GEMM 4096^3) in f32 / f64 / complex64 / complex128, the
                    dtypes of the reference's own tests (f32 runs on the bf16 cores via the exact 3 x bf16 split).
  mera           -- configs[4] shape on one GPU: binary-MERA layer energy at chi = 32 (68.7 GB intermediate).
  sliced_network -- the north-star scaling network (64-node random 3-regular graph, bond
                    D, bf16): bond-sliced greedy contraction, slices dealt over the N
                    ranks, ONE all-reduce of the scalar (strong scaling: fixed total work).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md:42


def parse_args():
  p = argparse.ArgumentParser()
  p.add_argument("--gpus", type=int, default=1)
  p.add_argument("--steps", type=int, default=5)
  p.add_argument("--warmup", type=int, default=2)
  p.add_argument("--bond", type=int, default=256, help="bond dimension D of the rank-4 nodes")
  p.add_argument("--layout", default="L0", choices=["L0", "L1"])
  p.add_argument("--svd-n", type=int, default=4096, help="side of the split_node matrix (0 = skip)")
  p.add_argument("--rr-bond", type=int, default=12,
                 help="bond dimension of the 64-node random-regular network (0 = skip)")
  p.add_argument("--rr-min-slices", type=int, default=64)
  p.add_argument("--no-cpu-baseline", action="store_true")
  p.add_argument("--no-sweep", action="store_true", help="skip the bond-dimension sweep rows")
  p.add_argument("--mera-chi", type=int, default=32, help="bond dimension of the MERA layer network (0 = skip)")
  p.add_argument("--fill", default="normal", choices=["normal", "zeros"],
                 help="operand fill (zeros shows the DVFS-inflated number; never the headline)")
  return p.parse_args()


def dist_setup(n_gpus):
  """torch.distributed is plumbing here: barrier + max-over-ranks of the time."""
  rank = int(os.environ.get("RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if world == 1 and not os.environ.get("TNH_BENCH_FORCE_DIST"):
    return rank, world, local, None   # (the env knob exercises the RCCL path on a 1-GPU box)
  os.environ.setdefault("MASTER_PORT", "29511")
  import torch  # pylint: disable=import-outside-toplevel
  import torch.distributed as dist  # pylint: disable=import-outside-toplevel
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  torch.cuda.set_device(local)
  dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                          device_id=torch.device("cuda", local))
  assert world == n_gpus, (world, n_gpus)
  return rank, world, local, dist


def sync_all(be, dist):
  be.synchronize()
  if dist is not None:
    import torch  # pylint: disable=import-outside-toplevel
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()


def make_nodes(ta, be, D, layout, seed, fill):
  if fill == "zeros":
    A = be.zeros((D,) * 4, dtype=ta.bfloat16)
    B = be.zeros((D,) * 4, dtype=ta.bfloat16)
  else:
    A = be.device_random((D,) * 4, dtype=ta.bfloat16, seed=2 * seed + 1, normal=True, a=0.0, b=1.0 / D)
    B = be.device_random((D,) * 4, dtype=ta.bfloat16, seed=2 * seed + 2, normal=True, a=0.0, b=1.0 / D)
  return A, B


def one_step(ta, be, A, B, layout):
  a, b = ta.Node(A, backend=be), ta.Node(B, backend=be)
  if layout == "L0":
    a[2] ^ b[0]  # pylint: disable=pointless-statement
    a[3] ^ b[1]  # pylint: disable=pointless-statement
  else:
    a[1] ^ b[2]  # pylint: disable=pointless-statement
    a[3] ^ b[0]  # pylint: disable=pointless-statement
  return ta.contract_between(a, b)


def cpu_baseline(layout):
  """NumPy oracle (port of numpy_backend.tensordot) on a bounded sample: same
  node layout, fp32 on bf16-rounded inputs, D chosen for ~10-30 s of CPU work."""
  from oracle import numpy_oracle as orc  # pylint: disable=import-outside-toplevel
  try:
    from threadpoolctl import threadpool_info  # pylint: disable=import-outside-toplevel
    threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
  except Exception:  # pylint: disable=broad-except
    threads = os.cpu_count() or 1
  axes = [[2, 3], [0, 1]] if layout == "L0" else [[1, 3], [2, 0]]
  rng = np.random.default_rng(2)

  def run(D):
    A = orc.round_bf16(rng.standard_normal((D,) * 4) / D)
    B = orc.round_bf16(rng.standard_normal((D,) * 4) / D)
    t0 = time.perf_counter()
    orc.tensordot(A, B, axes)
    return time.perf_counter() - t0

  run(32)  # warm-up (BLAS thread pool)
  t64 = run(64)
  flops64 = 2.0 * 64**6
  D = 64
  for cand in (96, 128, 160):
    est = 2.5 * t64 * (cand / 64.0)**6   # the D=64 probe runs from cache: larger sizes are ~2.5x slower per flop
    if est <= 30.0:
      D = cand
  t = run(D) if D != 64 else t64
  flops = 2.0 * float(D)**6
  return {"value": flops / t / 1e12, "unit": "TFLOP/s", "cores": int(threads), "kind": "port",
          "sample": f"oracle tensordot, layout {layout}, D={D} (GEMM {D*D}^3), fp32 on bf16-rounded inputs, "
                    f"{t:.2f} s; D=64 probe {flops64 / t64 / 1e12:.3f} TFLOP/s"}


def svd_bench(ta, be, n, k):
  """configs[2]: split_node of a rank-6 fp32 node reshaped n x n, keep k."""
  side = round(n ** (1.0 / 3.0))
  if side**3 != n:
    shape, left_axes, right_axes = (n, n), [0], [1]
  else:
    shape, left_axes, right_axes = (side,) * 6, [0, 1, 2], [3, 4, 5]
  x = be.device_random(shape, dtype=np.float32, seed=3, normal=True)
  node = ta.Node(x, backend=be)
  be.synchronize()
  t0 = time.perf_counter()
  left, right, trun = ta.split_node(node, [node[i] for i in left_axes], [node[i] for i in right_axes],
                                    max_singular_values=k)
  be.synchronize()
  t = time.perf_counter() - t0
  nbytes = 4 * (n * n + n * k + n + k * n)  # read A, write u_k, all s, vh_k
  return {"n": n, "k": k, "seconds": t, "gbps": nbytes / t / 1e9, "sweeps": be.last_svd_sweeps,
          "algorithmic_bytes": nbytes, "trunc_len": int(trun.shape[0]),
          "workload": f"split_node of a {shape} f32 node as {n}x{n}, max_singular_values={k} (second call; "
                      "first call warms the allocator)",
          "note": "block one-sided Jacobi: f32-MFMA gram/update + LDS eigensolver per block pair; bound by "
                  "MFMA flops and LDS latency over ~17 sweeps, not by the algorithmic-bytes HBM figure "
                  "(see DESIGN.md)"}


def svd_cpu_baseline(n_full):
  """The oracle's SVD (np.linalg.svd, the reference's decompositions.py:36) on a bounded sample."""
  from oracle import numpy_oracle as orc  # pylint: disable=import-outside-toplevel
  rng = np.random.default_rng(3)

  def run(size):
    x = rng.standard_normal((size, size)).astype(np.float32)
    t0 = time.perf_counter()
    orc.svd(x, 1, max_singular_values=size // 16)
    return time.perf_counter() - t0

  n = min(n_full, 1024)
  t = run(n)
  # largest power-of-two size up to the benchmark's own whose LAPACK time (~n^3) stays under ~25 s
  while 2 * n <= n_full and t * 8.0 <= 25.0:
    n *= 2
    t = run(n)
  k = n // 16
  nbytes = 4 * (n * n + n * k + n + k * n)
  return {"value": nbytes / t / 1e9, "unit": "GB/s", "seconds": t, "kind": "port",
          "sample": f"oracle svd (np.linalg.svd) of {n}x{n} f32, keep {k}"}


def sliced_network_bench(ta, be, dist, rank, world, D, min_slices):
  """64-node random 3-regular network, bf16, bond-sliced greedy contraction."""
  from tensornetwork_amd import distributed, workloads  # pylint: disable=import-outside-toplevel
  n = 64
  tensors = workloads.random_regular_device_tensors(be, n, D, ta.bfloat16, seed=6)
  nodes = workloads.random_regular_network(be, n=n, D=D, seed=6, tensors=tensors)
  cuts = distributed.choose_cut_edges(nodes, min_slices=min_slices)
  rep = distributed.slicing_report(nodes, cuts)
  comm = distributed.TorchDistComm() if dist is not None else distributed.LocalComm()
  # warm-up on a few slices (allocator, kernels), then the timed full contraction
  class _Few(distributed.LocalComm):
    rank, world = 0, max(1, int(rep["n_slices"]) // 2)
  distributed.contract_sliced(nodes, cuts, comm=_Few())
  sync_all(be, dist)
  t0 = time.perf_counter()
  out = distributed.contract_sliced(nodes, cuts, comm=comm)
  sync_all(be, dist)
  t = time.perf_counter() - t0
  if dist is not None:
    import torch  # pylint: disable=import-outside-toplevel
    tt = torch.tensor([t], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t = float(tt.item())
  total_flops = 2.0 * rep["flops_per_slice"] * rep["n_slices"]   # the cost model counts multiply-adds
  return {"workload": f"64-node random 3-regular network (seed 6), bond D={D}, bf16, {len(cuts)} cut bonds",
          "n_slices": int(rep["n_slices"]), "n_gpus": world, "seconds": t, "scaling": "strong",
          "flops_total": total_flops, "tflops": total_flops / t / 1e12,
          "peak_intermediate_elems": rep["peak_per_slice"],
          "collective": "one all-reduce(sum) of the scalar" if world > 1 else "none",
          "result": float(np.asarray(out).reshape(-1)[0])}


def bond_sweep(ta, be):
  """Whole-path TFLOP/s of contract_between over the bond-dimension sweep (rank 0, N = 1 only)."""
  rows = []
  for D in (32, 64, 96, 128, 192, 256):
    A, B = make_nodes(ta, be, D, "L0", seed=7, fill="normal")
    for layout in ("L0", "L1"):
      reps = 3 if D >= 192 else 10
      one_step(ta, be, A, B, layout)
      be.synchronize()
      t0 = time.perf_counter()
      for _ in range(reps):
        out = one_step(ta, be, A, B, layout)
        del out
      be.synchronize()
      t = (time.perf_counter() - t0) / reps
      rows.append({"D": D, "layout": layout, "gemm": [D * D] * 3, "ms": t * 1e3, "tflops": 2.0 * D**6 / t / 1e12,
                   "kernel": be.lib.tnh_gemm_last_kernel().decode()})
    del A, B
  # north-star row: two shared D = 512 bonds, M = N = 8192, K = 262144
  A = be.device_random((64, 128, 512, 512), dtype=ta.bfloat16, seed=11, normal=True, a=0.0, b=1.0 / 512)
  B = be.device_random((512, 512, 128, 64), dtype=ta.bfloat16, seed=12, normal=True, a=0.0, b=1.0 / 512)
  def step():
    a, b = ta.Node(A, backend=be), ta.Node(B, backend=be)
    a[2] ^ b[0]  # pylint: disable=pointless-statement
    a[3] ^ b[1]  # pylint: disable=pointless-statement
    return ta.contract_between(a, b)
  step()
  be.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    out = step()
    del out
  be.synchronize()
  t = (time.perf_counter() - t0) / 3
  rows.append({"D": 512, "layout": "A(64,128,512,512).B(512,512,128,64)", "gemm": [8192, 8192, 262144], "ms": t * 1e3,
               "tflops": 2.0 * 8192 * 8192 * 262144 / t / 1e12, "kernel": be.lib.tnh_gemm_last_kernel().decode()})
  return rows


def dtype_sweep(ta, be, D=64):
  """backend.tensordot(a, b, [[2, 3], [0, 1]]) -- the call contract_between makes for layout L0 -- on two
  rank-4 tensors (GEMM D^2 cubed) in the dtypes the reference's own tests use: f32 (large products run on
  the bf16 matrix cores from the exact 3 x bf16 split), f64, complex64 / complex128 (real-expansion GEMM).
  TFLOP/s counts real flops: 2 MNK, 8 MNK for complex."""
  rows = []
  for name, dt, mult in (("f32", np.float32, 2.0), ("f64", np.float64, 2.0), ("complex64", np.complex64, 8.0),
                         ("complex128", np.complex128, 8.0)):
    A = be.device_random((D,) * 4, dtype=dt, seed=21, normal=True, b=1.0 / D)
    B = be.device_random((D,) * 4, dtype=dt, seed=22, normal=True, b=1.0 / D)
    be.tensordot(A, B, [[2, 3], [0, 1]])
    be.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
      out = be.tensordot(A, B, [[2, 3], [0, 1]])
      del out
    be.synchronize()
    t = (time.perf_counter() - t0) / reps
    rows.append({"dtype": name, "D": D, "gemm": [D * D] * 3, "ms": t * 1e3, "tflops": mult * D**6 / t / 1e12,
                 "kernel": be.lib.tnh_gemm_last_kernel().decode()})
    del A, B
  return rows


def mera_bench(ta, be, chi):
  """configs[4] shape on one GPU: binary-MERA layer energy (12-node network, both placements,
  contractors.branch nbranch=2), bf16 operands generated in HBM; flops from the path cost model."""
  from tensornetwork_amd import contractors, network, pathfinder, workloads  # pylint: disable=import-outside-toplevel
  sc = lambda n: float(n) ** -0.5
  ham = be.device_random((chi,) * 6, dtype=ta.bfloat16, seed=1, normal=True, b=sc(chi**3))
  rho = be.device_random((chi,) * 6, dtype=ta.bfloat16, seed=2, normal=True, b=sc(chi**3))
  iso = be.device_random((chi,) * 3, dtype=ta.bfloat16, seed=3, normal=True, b=sc(chi))
  dis = be.device_random((chi,) * 4, dtype=ta.bfloat16, seed=4, normal=True, b=sc(chi * chi))
  nodes = workloads.mera_layer_network(be, ham, rho, iso, dis, "left")
  inputs = [set(n.edges) for n in nodes]
  sizes = {e: e.dimension for e in network.get_all_edges(nodes)}
  macs, peak = pathfinder.path_cost(inputs, set(), sizes, pathfinder.branch(inputs, set(), sizes, nbranch=2))
  del nodes
  run = lambda: workloads.mera_energy(be, ham, rho, iso, dis, lambda nd: contractors.branch(nd, nbranch=2))
  run()
  be.synchronize()
  t0 = time.perf_counter()
  out = run()
  be.synchronize()
  t = time.perf_counter() - t0
  return {"workload": f"binary-MERA layer energy, chi={chi}, bf16, left + right placement, contractors.branch(nbranch=2)",
          "seconds": t, "flops": 4.0 * float(macs), "tflops": 4.0 * float(macs) / t / 1e12,
          "peak_intermediate_elems": float(peak), "energy": float(np.asarray(out).reshape(-1)[0])}


def load_traffic(kernel_name, M, N, K):
  """HBM bytes per launch of the headline kernel from the committed rocprofv3 PMC pass."""
  import glob  # pylint: disable=import-outside-toplevel
  for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
    try:
      with open(path) as f:
        rec = json.load(f)
    except (OSError, ValueError):
      continue
    if rec.get("kernel") == kernel_name and rec.get("shape") == [M, N, K]:
      return rec.get("hbm_bytes"), os.path.basename(path)
  return None, None


def main():
  args = parse_args()
  rank, world, local, dist = dist_setup(args.gpus)
  os.environ.setdefault("TNHIP_DEVICE", str(local))
  import tensornetwork_amd as ta  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import _lib  # pylint: disable=import-outside-toplevel

  be = ta.get_hip_backend()
  D = args.bond
  M = N = K = D * D
  flops_per_step = 2.0 * M * N * K
  A, B = make_nodes(ta, be, D, args.layout, seed=rank, fill=args.fill)

  for _ in range(args.warmup):
    out = one_step(ta, be, A, B, args.layout)
    del out
  be.gemm_events = []          # HIP events around every GEMM launch in the timed region
  sync_all(be, dist)
  t0 = time.perf_counter()
  for _ in range(args.steps):
    out = one_step(ta, be, A, B, args.layout)
    del out
  sync_all(be, dist)
  elapsed = time.perf_counter() - t0
  events, be.gemm_events = be.gemm_events, None
  kernel_name = be.lib.tnh_gemm_last_kernel().decode()

  if dist is not None:
    import torch  # pylint: disable=import-outside-toplevel
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  gemm_ms = [s.elapsed_ms(e) for s, e in events]
  gemm_avg_s = (sum(gemm_ms) / len(gemm_ms) / 1e3) if gemm_ms else float("nan")
  achieved = flops_per_step / gemm_avg_s / 1e12 if gemm_ms else float("nan")

  traffic, traffic_src = load_traffic(kernel_name, M, N, K)
  result = {
      "metric": "contracted-elements/sec (TFLOP/s) + SVD GB/s, bond-dim sweep, 1/2/4/8 MI355X",
      "value": flops_per_step * args.steps * world / elapsed / 1e12,
      "unit": "TFLOP/s",
      "n_gpus": world,
      "steps": args.steps,
      "warmup": args.warmup,
      "ms_per_step": elapsed / args.steps * 1e3,
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": "bf16",
      "data": "synthetic" if args.fill == "normal" else "synthetic-zeros",
      "config": {"workload": f"contract_between, two rank-4 bf16 nodes, bond D={D}, layout {args.layout} "
                             f"(GEMM {M}x{N}x{K}, fp32 accumulate, bf16 out)",
                 "parallelism": "1 GPU" if world == 1 else f"{world} independent pairwise contractions "
                                                           "(no data-path collective)"},
      "roofline": {"bound": "mfma", "achieved": achieved, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": achieved / BF16_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                   "algorithmic_bytes": 2.0 * (M * K + N * K + M * N), "kernel": kernel_name,
                   "kernel_ms": gemm_avg_s * 1e3, "launches": len(gemm_ms)},
  }
  del A, B
  _lib.check(be.lib.tnh_trim())
  if args.rr_bond > 0:
    try:
      sliced = sliced_network_bench(ta, be, dist, rank, world, args.rr_bond, args.rr_min_slices)
    except Exception as exc:  # pylint: disable=broad-except
      sliced = {"error": f"{type(exc).__name__}: {exc}"}
    result["sliced_network"] = sliced
  if rank == 0:
    if world == 1 and not args.no_sweep:
      try:
        result["bond_sweep"] = bond_sweep(ta, be)
      except Exception as exc:  # pylint: disable=broad-except
        result["bond_sweep"] = {"error": f"{type(exc).__name__}: {exc}"}
      _lib.check(be.lib.tnh_trim())
      try:
        result["dtype_sweep"] = dtype_sweep(ta, be)
      except Exception as exc:  # pylint: disable=broad-except
        result["dtype_sweep"] = {"error": f"{type(exc).__name__}: {exc}"}
      _lib.check(be.lib.tnh_trim())
    if world == 1 and args.mera_chi > 0:
      try:
        result["mera"] = mera_bench(ta, be, args.mera_chi)
      except Exception as exc:  # pylint: disable=broad-except
        result["mera"] = {"error": f"{type(exc).__name__}: {exc}"}
      _lib.check(be.lib.tnh_trim())
    # every secondary leg is fenced: whatever happens in one of them, the headline line is printed
    if world == 1 and args.svd_n > 0:
      try:
        svd_bench(ta, be, args.svd_n, max(args.svd_n // 16, 1))  # warm-up
        result["svd"] = svd_bench(ta, be, args.svd_n, max(args.svd_n // 16, 1))
        if not args.no_cpu_baseline:
          result["svd"]["cpu_baseline"] = svd_cpu_baseline(args.svd_n)
      except Exception as exc:  # pylint: disable=broad-except
        result.setdefault("svd", {})["error"] = f"{type(exc).__name__}: {exc}"
    if world == 1 and not args.no_cpu_baseline:
      try:
        result["cpu_baseline"] = cpu_baseline(args.layout)
      except Exception as exc:  # pylint: disable=broad-except
        result["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    print(json.dumps(result))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
