"""Headline benchmark: ONE compact JSON line (the last line of stdout, < 4 KB: every contract key, `roofline`,
`cpu_baseline`, `verified` {all_ok, checks, failed}, one or two numbers per secondary leg) + the full record in
`bench_detail.json` next to this file and under gpurun_out/ (round 4: the one 21 KB line of round 3 did not fit the
driver's 8 KB tail).  The objects described below are those of the FULL record.

Workload at N=1 (BASELINE.json configs[1]): ``contract_between`` of two rank-4
bf16 nodes with two shared bonds of dimension D=256 (layout L0: a[2]^b[0],
a[3]^b[1]) -> one 65536^3 GEMM on the MFMA path, operands generated in HBM.
A "step" is one such contract_between through the product path
(Node bookkeeping -> HipBackend.tensordot -> ONE K1 pass over the k-major operand `b` -> tnh_gemm_view: the lean
ping-pong kernel on two K-contiguous views).
``value`` = 2*M*N*K*steps / wall time, TFLOP/s, inputs resident in HBM.

N>1 (one process per GPU; RCCL through libtnhip's own K8 entry points for the barrier, the
max-over-ranks reduction and the sliced network's all-reduce): the pairwise contraction has no
exchange step, so every rank contracts its own pair of nodes (weak scaling, no data-path
collective) and ``value`` is the aggregate.  If the RCCL communicator cannot be brought up (it then raises on every
rank, in step), the barrier and the scalar reductions go through the host rendezvous instead (`HostComm`, TCP) and
``config.comm`` says so: the headline is the same measurement, the sliced network's exchange is labelled a host one.

Extra objects on the same line:
  roofline       -- the dominant kernel (bf16 MFMA GEMM) timed with HIP events on the library's stream inside
                    the timed region, vs 2.5 PFLOP/s; board power and shader clock sampled from sysfs over the
                    same region give `frac_at_observed_clock`; `traffic` = HBM bytes per launch from the
                    committed rocprofv3 PMC pass of this same command (profiles/*_traffic.json), else null.
  verified       -- value checks of the BASELINE-size outputs, OUTSIDE the timed regions: 1024 sampled
                    entries of the D=256 L0 / L1 results and of the D=512 row vs float64 dot products of the
                    device operands; configs[2] at 4096^2 vs LAPACK (all 4096 singular values incl. s_rest,
                    orthonormality, reconstruction vs the best rank-k error); the bf16 sliced network vs an f32
                    run of the same tensors; the MERA layer at chi = 16 vs the float64 oracle.
  cpu_baseline   -- the NumPy oracle (port of the reference's tensordot; `kind: "port"`) timed on this box's host cores
                    on a bounded sample (same layout, smaller D) -- or, when the builder shipped the reference beside
                    the repo for the call (TN_REFERENCE_DIR), google/TensorNetwork's own contract_between on its NumPy
                    backend (`kind: "reference"`).
  svd            -- split_node truncated SVD (configs[2]: (16,)*6 node -> 4096 x 4096, keep 256) in the metric's
                    GB/s; `sweep` holds SURVEY 8d's full case list: Gaussian and s_i = 2^(-i/32) inputs, natural and
                    mixed edge order, n = 512 .. 4096, each checked against LAPACK (verified.svd), plus one row per
                    call shape of round 4 (truncation error alone, full SVD, a side that is not a multiple of 16,
                    float64: verified.svd_call_shapes); `bound` = what really bounds the band SVD (dependent
                    launches, rank-16 update traffic, f64 Sturm counts); the oracle's LAPACK SVD timed on a bounded
                    sample beside it.
  bond_sweep     -- the metric's bond-dimension sweep: contract_between of two rank-4 bf16 nodes at
                    D = 32 .. 256 in the favourable (L0) and the permute-needing (L1) layout, plus the
                    north-star "D = 512" row A(64,128,512,512) . B(512,512,128,64) (GEMM 8192 x 8192 x 262144),
                    whole-path TFLOP/s each, with the K1 permute launches per contraction.
  dtype_sweep    -- the same contraction at D = 64 in f32 / f64 / complex64 / complex128.
  mps_chain      -- configs[3]: <psi|psi> of a 16-site MPS, bulk D = 512, contractors.greedy (d = 2 and d = 4;
                    eager and hipGraph replay) with the NumPy oracle backend's time beside it.
  mera           -- configs[4] shape on one GPU: binary-MERA layer energy at chi = 32 (68.7 GB intermediate).
  mera_chi64     -- configs[4] at chi = 64: `--mera64-full` placements (default 1 of 2) run in full -- all 4096 slices that
                    `slice_edge` on one leg of the hamiltonian and one leg of the state leaves (BOTH tensors on a cut
                    leg sliced), through the machinery of contract_sliced; a step runs once per distinct value of the
                    cuts it depends on: `measured_*` with the EXECUTED flops = the dense-optimal cost of the network
                    -- the rest as per-slice cost x slice count (labelled extrapolated: one slice alone x 4096).
  helpers        -- HBM-bound helper kernels (K1 permute, K3/K4 reductions, K5 scaling) in GB/s vs 8 TB/s.
  gather_gemm    -- one product of the D = 12 network (a 144 x 144 tensor takes two bonds off a 430 M-element rank-8
                    intermediate) for five placements of the contracted axes + the K = 1728 product: `tnh_gemm_gather`
                    against permute + streaming GEMM, both operand orders, device-side equality check.
  sliced_network -- the north-star scaling network (64-node random 3-regular graph, bond D = 16 from round 5 on: the 16
                    values of the costly cut divide evenly among 2 / 4 / 8 ranks; bf16): bond-sliced greedy contraction,
                    slices dealt over the N ranks (`distributed._StagePlan.partition`), ONE all-reduce of the scalar
                    (strong scaling: fixed total work); per-rank compute and all-reduce times.  Every step of the path
                    runs once per distinct value of the cut bonds it depends on (`mode`); `tflops` counts the executed
                    flops as the run itself counted them (summed over ranks), `flops_if_every_slice_ran_alone` is what
                    the stand-alone slices would cost.  `sliced_network_small`: the D = 12 instance of rounds 1-4 (N = 1).
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
F32_MFMA_PEAK_TFLOPS = 157.3  # dense f32 matrix peak, same table
F64_MFMA_PEAK_TFLOPS = 78.6   # f64 matrix peak (spec)
HBM_PEAK_GBPS = 8000.0          # spec, MI355X_MICROARCH.md:35 (about 6.3 TB/s achievable)
MAX_CLOCK_MHZ = 2400.0


def parse_args(argv=None):
  p = argparse.ArgumentParser()
  p.add_argument("--gpus", type=int, default=1)
  p.add_argument("--steps", type=int, default=5)
  p.add_argument("--warmup", type=int, default=2)
  p.add_argument("--bond", type=int, default=256, help="bond dimension D of the rank-4 nodes")
  p.add_argument("--layout", default="L0", choices=["L0", "L1"])
  p.add_argument("--svd-n", type=int, default=4096, help="side of the split_node matrix (0 = skip)")
  p.add_argument("--rr-bond", type=int, default=16,
                 help="bond dimension of the 64-node random-regular network (0 = skip).  16 (round 5; rounds 1-4: 12): "
                      "the 16 values of the costly cut bond divide evenly among 2 / 4 / 8 ranks (12 values on 8 ranks "
                      "cap the speed-up at 6.0x before any overhead) and one contraction is 1.3e15 flop, so per-rank "
                      "fixed costs no longer show")
  p.add_argument("--rr-bond-small", type=int, default=12,
                 help="second, smaller instance of the same network, one GPU only (the rounds 1-4 row; 0 = skip)")
  p.add_argument("--rr-min-slices", type=int, default=64)
  p.add_argument("--no-cpu-baseline", action="store_true")
  p.add_argument("--no-sweep", action="store_true", help="skip the bond-dimension sweep rows")
  p.add_argument("--no-verify", action="store_true", help="skip the value checks (verified object)")
  p.add_argument("--no-extras", action="store_true", help="skip mps_chain / mera_chi64 / helpers")
  p.add_argument("--mera-chi", type=int, default=32, help="bond dimension of the MERA layer network (0 = skip)")
  p.add_argument("--mera64-full", type=int, default=1, choices=[0, 1, 2],
                 help="placements of the chi = 64 MERA layer run in FULL (4096 slices each, ~150 s per placement on one "
                      "MI355X); the others are reported per slice x count")
  p.add_argument("--mera64-budget", type=float, default=150.0, help="seconds after which a full placement stops early")
  p.add_argument("--bringup-timeout", type=float, default=float(os.environ.get("TNH_BENCH_BRINGUP_TIMEOUT_S", "900")),
                 help="N > 1: seconds the ranks get to rendezvous, create the RCCL communicator and pass the first "
                      "barrier before the job is killed with a message (a hang inside RCCL must not become the record)")
  p.add_argument("--job-timeout", type=float, default=float(os.environ.get("TNH_BENCH_JOB_TIMEOUT_S", "2700")),
                 help="self-launched N > 1 jobs: seconds before the launcher kills every rank")
  p.add_argument("--fill", default="normal", choices=["normal", "zeros"],
                 help="operand fill (zeros shows the DVFS-inflated number; never the headline)")
  p.add_argument("--allow-host-exchange", action="store_true",
                 help="N > 1 only: when the RCCL communicator does not come up, carry on with the barrier / scalar reductions "
                      "over the host rendezvous (TCP) and say so in config.comm; WITHOUT this flag that failure ends the "
                      "run with a non-zero exit code -- a SCALE record must never show N ranks and no RCCL by accident")
  p.add_argument("--dry-run", action="store_true",
                 help="launcher / rendezvous check only: every rank joins the host rendezvous, rank 0 prints a JSON "
                      "line with n_gpus; no GPU is touched (CPU test of the --gpus N self-launch)")
  return p.parse_args(argv)


# --------------------------------------------------------------------------- launcher
def visible_gpus():
  """Devices the HIP runtime shows this process (0 when there is none or the library cannot ask)."""
  import ctypes  # pylint: disable=import-outside-toplevel
  try:
    from tensornetwork_amd import _lib  # pylint: disable=import-outside-toplevel
    n = ctypes.c_int(0)
    if _lib.load_library().tnh_device_count(ctypes.byref(n)) != 0:
      return 0
    return int(n.value)
  except Exception:  # pylint: disable=broad-except
    return 0


def self_launch(args):
  """`python bench.py --gpus N` with N > 1 and no launcher environment: be the launcher.

  One rank per GPU, torchrun's environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT on
  loopback), rank 0's stdout is this process' stdout (the ONE JSON line), the other ranks' stdout goes to stderr.
  Fewer visible devices than ranks is an error, never a silent single-GPU run.  Returns the exit code."""
  import socket  # pylint: disable=import-outside-toplevel
  import subprocess  # pylint: disable=import-outside-toplevel
  n = args.gpus
  if not args.dry_run:
    have = visible_gpus()
    if have < n:
      print(f"[bench] --gpus {n} asked for {n} ranks (one per GPU) but this process sees {have} device(s); "
            "refusing to report a smaller job under that flag", file=sys.stderr, flush=True)
      return 2
  with socket.socket() as sock:
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                   "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                  stdout=None if r == 0 else sys.stderr))
  # Watch the ranks instead of waiting for them one by one: the first rank that fails takes the others down (they
  # would otherwise sit in a collective for ever), and the whole job has a wall-clock limit.
  deadline = time.monotonic() + args.job_timeout
  codes = [None] * n
  why = None
  while any(c is None for c in codes):
    for r, proc in enumerate(procs):
      if codes[r] is None:
        codes[r] = proc.poll()
    failed = {r: c for r, c in enumerate(codes) if c not in (None, 0)}
    if failed:
      why = f"rank exit codes {failed}"
    elif time.monotonic() > deadline:
      why = f"job exceeded --job-timeout {args.job_timeout:.0f} s"
    if why:
      for r, proc in enumerate(procs):
        if codes[r] is None:
          proc.kill()
          codes[r] = proc.wait()
      break
    time.sleep(0.05)
  if why:
    print(f"[bench] {why}; every rank stopped, no result line", file=sys.stderr, flush=True)
    return 1
  return 0


def dry_run(args, rank, world):
  """Rendezvous only (no GPU): proves that N ranks were started, found each other and agree on the world size."""
  from tensornetwork_amd import comm as tcomm  # pylint: disable=import-outside-toplevel
  rdv = tcomm.HostRendezvous(rank, world, timeout=120)
  seen = rdv.all_gather({"rank": rank, "pid": os.getpid(), "local": int(os.environ.get("LOCAL_RANK", "0"))})
  rdv.barrier()
  rdv.close()
  if rank == 0:
    print(json.dumps({"metric": "contracted-elements/sec (TFLOP/s) + SVD GB/s, bond-dim sweep, 1/2/4/8 MI355X",
                      "dry_run": True, "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                      "ranks_seen": sorted(x["rank"] for x in seen),
                      "distinct_processes": len({x["pid"] for x in seen})}), flush=True)


# --------------------------------------------------------------------------- communicators
class Watchdog:
  """Bounded time for a step that can hang inside a collective (communicator bring-up, the first barrier): if the
  block does not finish within `seconds`, say so on stderr and end THIS process with exit code 3 -- under torchrun
  or `self_launch` the failing rank takes the job down, so a hang becomes a failed run with a message instead of a
  driver-side timeout.  ctypes releases the GIL during library calls, so the timer thread runs while the main
  thread is stuck in RCCL; `os._exit` because a normal exit would wait for the stuck call."""

  def __init__(self, seconds, what, rank):
    import threading  # pylint: disable=import-outside-toplevel
    self._timer = threading.Timer(seconds, self._fire) if seconds and seconds > 0 else None
    self._what, self._rank, self._seconds = what, rank, seconds
    if self._timer is not None:
      self._timer.daemon = True

  def _fire(self):
    print(f"[bench] rank {self._rank}: {self._what} did not finish within {self._seconds:.0f} s "
          "(--bringup-timeout); stopping the job", file=sys.stderr, flush=True)
    os._exit(3)  # pylint: disable=protected-access

  def __enter__(self):
    if self._timer is not None:
      self._timer.start()
    return self

  def __exit__(self, *exc):
    if self._timer is not None:
      self._timer.cancel()
    return False


def make_rccl_comm(tcomm, be, rank, world):
  """K8 communicator (RCCL through libtnhip's C ABI).  A rank that cannot create it raises on EVERY rank (lock-step
  bootstrap, comm.py), so the ranks can agree on what to do next in step (`HostComm` below)."""
  return tcomm.RcclComm(be, rank=rank, world=world)


class HostComm:
  """bench.py only, and only when the K8 communicator cannot be brought up on this node (the exception of
  `make_rccl_comm`, raised on every rank): the barrier, the max / sum of a scalar over the ranks and the ONE small
  all-reduce of the sliced network go through `comm.HostRendezvous` (TCP on MASTER_ADDR) instead of RCCL, and the
  result line says so (`config.comm`).  The headline needs no data-path collective -- N independent contractions
   -- so its number is the same measurement; a row whose exchange is the thing measured
  (`sliced_network.allreduce_seconds`) is labelled as a host exchange.  Never used by the package."""

  MAX_ELEMS = 1 << 20           # the result of a sliced contraction is small (a scalar for closed networks)

  def __init__(self, rdv, reason):
    self._rdv, self.reason = rdv, str(reason)
    self.rank, self.world = rdv.rank, rdv.world

  def barrier(self):
    self._rdv.barrier()

  def max_over_ranks(self, value):
    return max(float(v) for v in self._rdv.all_gather(float(value)))

  def sum_over_ranks(self, value):
    return float(sum(float(v) for v in self._rdv.all_gather(float(value))))

  def all_gather_counts(self, n):
    return [int(x) for x in self._rdv.all_gather(int(n))]

  def all_reduce_sum(self, backend, tensor):
    """Sum over ranks on the host: D2H, gathered as lists, added in float64 in rank order on every rank (so every
    rank holds the same bits), rounded once to the tensor's dtype, H2D."""
    from tensornetwork_amd.device_tensor import DeviceTensor, bfloat16  # pylint: disable=import-outside-toplevel
    host = np.asarray(tensor)
    if host.size > self.MAX_ELEMS:
      raise RuntimeError(f"HostComm.all_reduce_sum is for small results (<= {self.MAX_ELEMS} elements), got {host.size}")
    cplx = np.iscomplexobj(host)
    flat = host.reshape(-1)
    mine = [flat.real.astype(np.float64).tolist(), flat.imag.astype(np.float64).tolist() if cplx else None]
    total_re, total_im = np.zeros(flat.size), np.zeros(flat.size)
    for re, im in self._rdv.all_gather(mine):
      total_re += np.asarray(re, dtype=np.float64)
      if im is not None:
        total_im += np.asarray(im, dtype=np.float64)
    total = (total_re + 1j * total_im) if cplx else total_re
    is_bf16 = getattr(tensor, "dtype", None) is bfloat16 or str(getattr(tensor, "dtype", "")) == "bfloat16"
    out = total.astype(np.float32 if is_bf16 else host.dtype).reshape(host.shape)
    if isinstance(tensor, DeviceTensor):
      return DeviceTensor.from_numpy(out, dtype=bfloat16) if is_bf16 else DeviceTensor.from_numpy(out)
    return out

  def close(self):
    self._rdv.close()


RCCL_COMM_NAME = "libtnhip K8 (tnh_allreduce / tnh_allgather over RCCL), TCP rendezvous for the id"
EXIT_NO_RCCL = 3       # exit code of a multi-rank run whose RCCL communicator did not come up (no --allow-host-exchange)


def bring_up_comm(tcomm, be, rank, world, allow_host_exchange=False):
  """(communicator, its name for `config.comm`): RCCL through the C ABI.  If that raises (on every rank, in step):
  the run ENDS with exit code EXIT_NO_RCCL -- unless `allow_host_exchange`, in which case the host exchange takes
  over and every field that names the communicator says so."""
  try:
    comm = make_rccl_comm(tcomm, be, rank, world)
    return comm, RCCL_COMM_NAME
  except (RuntimeError, TimeoutError) as exc:          # raised on every rank, in step
    reason = f"{type(exc).__name__}: {exc}"[:240]
    if not allow_host_exchange:
      print(f"[bench] rank {rank}: the RCCL communicator did not come up ({reason}); a {world}-rank line without RCCL is "
            "not a measurement of this framework's multi-GPU path -- exiting (pass --allow-host-exchange to run the "
            "headline over the host rendezvous anyway, labelled)", file=sys.stderr, flush=True)
      raise SystemExit(EXIT_NO_RCCL) from exc
    print(f"[bench] rank {rank}: the RCCL communicator did not come up ({reason}); barrier / scalar reductions go "
          "through the host rendezvous (TCP) instead -- the headline has no data-path collective", file=sys.stderr, flush=True)
    base = int(os.environ.get("MASTER_PORT", "29500")) + int(os.environ.get("TNH_COMM_PORT_OFFSET", "23"))
    rdv = tcomm.HostRendezvous(rank, world, port=base + 1, timeout=120)
    return HostComm(rdv, reason), f"HOST EXCHANGE (TCP rendezvous): the RCCL communicator did not come up -- {reason}"


def rccl_ranks(be, comm):
  """World size of the LIVE RCCL communicator as the library reports it (tnh_comm_info), 0 when the exchange is not
  RCCL (one rank without a communicator, or the labelled host exchange)."""
  import ctypes  # pylint: disable=import-outside-toplevel
  if comm is None or isinstance(comm, HostComm):
    return 0
  r, w = ctypes.c_int(-1), ctypes.c_int(0)
  if be.lib.tnh_comm_info(ctypes.byref(r), ctypes.byref(w)) != 0:
    return 0
  return int(w.value)


def sync_all(be, comm):
  be.synchronize()
  if comm is not None:
    comm.barrier()        # device-level: a 1-element all-reduce on the library stream, then a stream sync
    be.synchronize()


# --------------------------------------------------------------------------- headline pieces
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


def verify_pair(*args, **kwargs):
  """tests/cases.py::verify_pair (shared with the `-m gpu` suite, which runs the same check at the BASELINE sizes):
  >= 1024 sampled entries against float64 dot products of the device operands, 2^-8 |ref| + 2^-10 rms(ref)."""
  tdir = os.path.join(ROOT, "tests")
  if tdir not in sys.path:
    sys.path.insert(0, tdir)
  import cases  # pylint: disable=import-outside-toplevel,import-error
  return cases.verify_pair(*args, **kwargs)


def host64(t):
  return np.asarray(t).astype(np.float64)


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
  # The reference itself, when the builder shipped it beside the repo for this call (TN_REFERENCE_DIR, the scratch
  # copy of tools/reference_dropin/gpurun_with_reference.sh; the driver's box has none): its own contract_between on
  # its own NumPy backend -> kind "reference".  Otherwise the oracle -> kind "port".
  tn_ref = None
  ref_dir = os.environ.get("TN_REFERENCE_DIR")
  if ref_dir and os.path.isdir(os.path.join(ref_dir, "tensornetwork")):
    try:
      # inert stubs for what the image lacks (h5py, graphviz, jax / tensorflow; opt_einsum -> our path finder): the
      # same ones the golden-fixture generators and the drop-in harness use
      for stub in ("tests/golden/_stubs", "tools/reference_dropin/_stubs"):
        sys.path.insert(0, os.path.join(ROOT, stub))
      sys.path.insert(0, ref_dir)
      import tensornetwork as tn_ref  # pylint: disable=import-outside-toplevel,import-error
    except Exception:  # pylint: disable=broad-except
      tn_ref = None

  def run(D):
    A = orc.round_bf16(rng.standard_normal((D,) * 4) / D)
    B = orc.round_bf16(rng.standard_normal((D,) * 4) / D)
    t0 = time.perf_counter()
    if tn_ref is not None:
      a, b = tn_ref.Node(A, backend="numpy"), tn_ref.Node(B, backend="numpy")
      for x, y in zip(*axes):
        a[x] ^ b[y]  # pylint: disable=pointless-statement
      tn_ref.contract_between(a, b)
    else:
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
  if tn_ref is not None:
    return {"value": flops / t / 1e12, "unit": "TFLOP/s", "cores": int(threads), "kind": "reference",
            "sample": f"google/TensorNetwork {getattr(tn_ref, '__version__', '?')} itself (tn.contract_between on its NumPy "
                      f"backend, shipped beside the repo for this call), layout {layout}, D={D} (GEMM {D*D}^3), fp32 on "
                      f"bf16-rounded inputs, {t:.2f} s; D=64 probe {flops64 / t64 / 1e12:.3f} TFLOP/s"}
  return {"value": flops / t / 1e12, "unit": "TFLOP/s", "cores": int(threads), "kind": "port",
          "sample": f"oracle tensordot (NumPy restatement of the reference's numpy_backend.tensordot; the reference "
                    f"itself is not installed on this box), layout {layout}, D={D} (GEMM {D*D}^3), fp32 on "
                    f"bf16-rounded inputs, {t:.2f} s; D=64 probe {flops64 / t64 / 1e12:.3f} TFLOP/s; the D=256 "
                    f"headline shape would take ~{2.0 * 256.0**6 / (flops / t) / 60.0:.0f} min at this rate (extrapolated)"}


# --------------------------------------------------------------------------- configs[2]
_SVD_DIMS = {512: (8, 8, 8), 1024: (8, 8, 16), 2048: (8, 16, 16), 4096: (16, 16, 16)}


def svd_case_matrix(be, n, kind, seed):
  """The n x n f32 matrix of one configs[2] case, made in HBM.  'gauss': N(0, 1) entries; 'graded': prescribed
  spectrum s_i = 2^(-i/32) between two orthogonal factors (the construction of decompositions_test.py:55-66, with
  the factors taken from the device QR of Gaussian matrices instead of a host SVD)."""
  if kind == "gauss":
    return be.device_random((n, n), dtype=np.float32, seed=seed, normal=True)
  qu, _ = be.qr(be.device_random((n, n), dtype=np.float32, seed=seed + 1, normal=True), 1)
  qv, _ = be.qr(be.device_random((n, n), dtype=np.float32, seed=seed + 2, normal=True), 1)
  spec = be.convert_to_tensor((2.0 ** (-np.arange(n) / 32.0)).astype(np.float32))
  return be.tensordot(be.broadcast_right_multiplication(qu, spec), qv, [[1], [1]])


def svd_case(ta, be, mat, n, k, order):
  """split_node of the rank-6 node whose (left | right) matricisation is `mat`; order 'natural': node axes
  (l0 l1 l2 r0 r1 r2), left edges [0, 1, 2]; 'mixed' (split_node_test.py:36-47): node axes (l0 r0 l1 r1 l2 r2),
  left edges [0, 2, 4], right edges [1, 3, 5] -- split_node has to permute before it can reshape."""
  d = _SVD_DIMS.get(n)
  if d is None:
    x, left_axes, right_axes = mat, [0], [1]
  elif order == "natural":
    x, left_axes, right_axes = be.reshape(mat, d + d), [0, 1, 2], [3, 4, 5]
  else:
    x, left_axes, right_axes = be.transpose(be.reshape(mat, d + d), (0, 3, 1, 4, 2, 5)), [0, 2, 4], [1, 3, 5]
  best, out, samples = float("inf"), None, []
  for rep in range(4):          # the first call warms the allocator; best of the next three (all listed in `samples_ms`:
    node = ta.Node(x, backend=be)      # ~2500 dependent launches of 5-30 us each follow the shader clock closely, and
    be.synchronize()                   # the first call after the 1400 W GEMM legs runs 10 % slow)
    t0 = time.perf_counter()
    left, right, trun = ta.split_node(node, [node[i] for i in left_axes], [node[i] for i in right_axes],
                                      max_singular_values=k)
    be.synchronize()
    t = time.perf_counter() - t0
    if rep >= 1:
      samples.append(t * 1e3)
      if t < best:
        best, out = t, (left.tensor, right.tensor, trun)
  nbytes = 4 * (n * n + n * k + n + k * n)  # read A, write u_k, all s, vh_k
  rec = {"n": n, "k": k, "order": order, "seconds": best, "gbps": nbytes / best / 1e9,
         "hbm_roofline_frac": nbytes / best / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": nbytes,
         "path": getattr(be, "last_svd_path", None), "sweeps": be.last_svd_sweeps, "samples_ms": samples}
  return rec, out


def check_svd_case(mat_host, s_ref, n, k, outputs):
  """One case against LAPACK (SURVEY 8c tolerances): split_node returns left = u sqrt(s), right = sqrt(s) vh
  (network_operations.py:219-226) and the discarded values.  Checked: ALL n singular values (kept ones through the
  column norms of `left`), orthonormality of the kept vectors, reconstruction against the best rank-k error."""
  left, right, trun = outputs
  lw = host64(left).reshape(n, k)          # u sqrt(s)
  rw = host64(right).reshape(k, n)         # sqrt(s) vh
  s_rest = host64(trun).reshape(-1)
  s_kept = np.sum(lw * lw, axis=0)         # column norms^2 of u sqrt(s) = s_i
  s_all = np.concatenate([s_kept, s_rest])
  s_err = float(np.max(np.abs(s_all - s_ref)) / s_ref[0])
  u = lw / np.sqrt(s_kept)[None, :]
  vh = rw / np.sqrt(s_kept)[:, None]
  orth_u = float(np.max(np.abs(u.T @ u - np.eye(k))))
  orth_v = float(np.max(np.abs(vh @ vh.T - np.eye(k))))
  recon = float(np.linalg.norm(mat_host - lw @ rw))
  best = float(np.sqrt(np.sum(s_ref[k:] ** 2)))
  rec_excess = (recon - best) / float(np.linalg.norm(mat_host))
  ok = s_err <= 1e-5 and orth_u <= 1e-4 and orth_v <= 1e-4 and abs(rec_excess) <= 1e-4 and len(s_rest) == n - k
  return {"s_max_err_over_s0": s_err, "s_rest_len": int(len(s_rest)), "orth_u": orth_u, "orth_vh": orth_v,
          "recon_minus_best_over_normA": rec_excess, "ok": bool(ok)}


def svd_bound(m, n, k, itemsize=4):
  """What bounds the band SVD (tnh_svd_band.hip) of an m x n matrix keeping k triplets -- NOT the metric's
  algorithmic bytes (75.5 MB at 4096^2: 9 us of HBM time).  Counted from the algorithm:
    launches       dependent kernel launches of stage 1: f32 input (round 6, tnh_svd_band_fast.inc) FOUR per pair of
                   16-wide panels -- two fused update + raw-pass sweeps with the panel factor as an extra workgroup,
                   two reduce kernels -- and the ten-launch loop of rounds 3-5 for the last 8 panels; f64 input that
                   loop throughout (14 per pair: two Cholesky-QR passes); ~40 for the spectrum slicing, ~12 for the
                   vectors;  floor = 4.5 us each back to back;
    update_bytes   f32: every fused sweep reads and writes the trailing block once, two sweeps per pair:
                   4 x itemsize x sum_p (m - 16 p)(n - 16 p); f64 (and f32 before round 6): a read for W = V^T C plus
                   a read + write for C -= V W, twice: 6 x;  floor at the 6.3 TB/s a copy reaches on this part;
    sturm_fma      f64 FMAs of the Sturm counts: 139 per pivot, n pivots per shift, 65536 + 15 n shifts for all values
                   (f64 input: three more rounds) + 3 x 15 k for the kept ones; this chip issues one 64-lane v_fma_f64
                   per SIMD every 5 cycles (measured, tools/fma_probe: 5.0-5.5 at one wave per SIMD; rounds 3-5 assumed
                   8): floor = FMAs / 64 x 5 cycles / (1024 SIMDs x 2.4 GHz).
  The three floors add up (the stages are dependent): that sum is the `floor_s` the measured time is compared with."""
  npanels = n // 16
  f64 = itemsize == 8
  slow_panels = npanels if f64 else min(npanels, 8)
  launches = (14 if f64 else 10) * slow_panels + 4 * (npanels - slow_panels) + 40 + (12 if f64 else 0) + 12
  tail = sum((m - 16 * p) * (n - 16 * p) for p in range(npanels))
  update_bytes = (6.0 if f64 else 4.0) * itemsize * tail
  rounds_all = 1 + (3 if f64 else 0)
  shifts = 65536 + rounds_all * 15 * n + (6 if f64 else 3) * 15 * k
  sturm_fma = 139.0 * n * shifts
  launch_floor = launches * 4.5e-6
  update_floor = update_bytes / 6.3e12
  sturm_floor = sturm_fma / 64.0 * 5.0 / (1024 * 2.4e9)
  return {"launches": launches, "launch_floor_s": launch_floor, "update_bytes": update_bytes,
          "update_floor_s": update_floor, "sturm_f64_fma": sturm_fma, "sturm_floor_s": sturm_floor,
          "floor_s": launch_floor + update_floor + sturm_floor,
          "note": "dependent-launch floor + rank-16 update traffic at copy rate + f64 Sturm counts at the vector pipe's "
                  "issue rate; the metric's algorithmic bytes would take microseconds"}


def svd_sweep(ta, be, n_max, verify):
  """configs[2] in full: inputs (i) Gaussian and (ii) s_i = 2^(-i/32), natural and mixed edge order,
  n in {512, 1024, 2048, 4096} up to n_max, keep n / 16; each case timed (second call) in seconds and the metric's
  GB/s, and checked against np.linalg.svd of the same matrix.  Returns (headline record = Gaussian, natural, n_max;
  all rows; the verified object)."""
  rows, checks = [], {}
  headline = None
  t_lapack = 0.0
  for n in [x for x in (512, 1024, 2048, 4096) if x <= n_max] or [n_max]:
    k = max(n // 16, 1)
    for kind in ("gauss", "graded"):
      mat = svd_case_matrix(be, n, kind, seed=3 + n)
      s_ref = mat_host = None
      if verify:
        mat_host = host64(mat).reshape(n, n)
        t0 = time.perf_counter()
        s_ref = np.linalg.svd(mat_host, compute_uv=False)
        t_lapack += time.perf_counter() - t0
        _S_REF[(kind, n)] = s_ref          # the call-shape rows below re-use the same matrices
      for order in ("natural", "mixed"):
        rec, out = svd_case(ta, be, mat, n, k, order)
        rec["input"] = kind
        if verify:
          chk = check_svd_case(mat_host, s_ref, n, k, out)
          rec["check"] = chk
          checks[f"{kind}_{order}_{n}"] = chk["ok"]
        rows.append(rec)
        if kind == "gauss" and order == "natural":
          headline = dict(rec)
        del out
      del mat
    be.lib.tnh_trim()
  headline = dict(headline or rows[-1])
  headline["workload"] = (f"split_node of a rank-6 f32 node as {headline['n']}x{headline['n']}, "
                          f"max_singular_values={headline['k']} (Gaussian, natural order; second call)")
  headline["bound"] = svd_bound(headline["n"], headline["n"], headline["k"])
  headline["frac_of_floor"] = headline["bound"]["floor_s"] / headline["seconds"]
  headline["note"] = ("min(m, n) >= 1024: band reduction (Cholesky-QR panels, rank-16 streaming updates) + spectrum "
                      "slicing on T = B^T B in f64 + inverse iteration + back-transformation (tnh_svd_band.hip); "
                      "smaller: block one-sided Jacobi.  Bound by dependent small kernels and f64 VALU, not by the "
                      "algorithmic-bytes HBM figure (DESIGN.md section 6)")
  verified = None
  if verify:
    worst = max(rows, key=lambda r: r["check"]["s_max_err_over_s0"])
    verified = {"cases": checks, "n_cases": len(checks), "worst_s_err_over_s0": worst["check"]["s_max_err_over_s0"],
                "worst_orth": max(max(r["check"]["orth_u"], r["check"]["orth_vh"]) for r in rows),
                "worst_recon_minus_best_over_normA": max(abs(r["check"]["recon_minus_best_over_normA"]) for r in rows),
                "lapack_values_only_seconds": t_lapack,
                "tol": "|s - s_lapack| <= 1e-5 s0 (all n values), orthonormality <= 1e-4, "
                       "(||A - L R||_F - best rank-k) <= 1e-4 ||A||_F",
                "ok": bool(all(checks.values()))}
  return headline, rows, verified


_S_REF = {}


def svd_wide_rows(ta, be, n_max, verify):
  """The call shapes beyond `max_singular_values` alone, one row each at the bench's size:
  truncation error alone (k picked on the host from the values), the full SVD of split_node_full_svd, a side that is
  not a multiple of the 16-wide panels, and float64 -- each timed like the sweep's rows (best of three after a warm-up
  call) and checked against LAPACK on ALL values plus the residual |A v - s u| of the kept triplets."""
  rows = []
  n = min(n_max, 4096)

  def run(mode, kind, mat, call, dtype="f32", tol=1e-5, tol_rest=None):
    best, out, samples = float("inf"), None, []
    for rep in range(4):
      be.synchronize()
      t0 = time.perf_counter()
      res = call(mat)
      be.synchronize()
      t = time.perf_counter() - t0
      if rep >= 1:
        samples.append(t * 1e3)
        if t < best:
          best, out = t, res
    u, sv, vh, rest = out
    m_, n_ = mat.shape
    k = int(sv.shape[0])
    item = 8 if dtype == "f64" else 4
    nbytes = item * (m_ * n_ + m_ * k + min(m_, n_) + k * n_)
    rec = {"n": int(min(m_, n_)), "k": k, "mode": mode, "dtype": dtype, "order": "natural", "input": kind,
           "seconds": best, "gbps": nbytes / best / 1e9, "algorithmic_bytes": nbytes, "samples_ms": samples,
           "path": getattr(be, "last_svd_path", None)}
    if verify:
      a = host64(mat)
      s_ref = _S_REF.get((kind, m_)) if m_ == n_ else None
      if s_ref is None:
        s_ref = np.linalg.svd(a, compute_uv=False)
      s_all = np.concatenate([host64(sv).reshape(-1), host64(rest).reshape(-1)])
      uu, vv = host64(u).reshape(m_, k), host64(vh).reshape(k, n_)
      chk = {"s_max_err_over_s0": float(np.max(np.abs(s_all - s_ref)) / s_ref[0]),
             "s_kept_err_over_s0": float(np.max(np.abs(s_all[:k] - s_ref[:k])) / s_ref[0]),
             "orth_u": float(np.max(np.abs(uu.T @ uu - np.eye(k)))), "orth_vh": float(np.max(np.abs(vv @ vv.T - np.eye(k)))),
             "triplet_resid_over_s0": float(np.max(np.linalg.norm(a @ vv.T - uu * s_all[:k], axis=0)) / s_ref[0])}
      # kept values to `tol`, discarded ones to `tol_rest` (f64 band path: 32-bit brackets and the sqrt(eps64) floor of
      # T = B^T B), orthonormality 10 tol, triplet residual 4 tol
      chk["tol"] = {"kept": tol, "rest": tol_rest if tol_rest is not None else tol}
      chk["ok"] = bool(chk["s_kept_err_over_s0"] <= tol and chk["s_max_err_over_s0"] <= chk["tol"]["rest"]
                       and chk["orth_u"] <= 10 * tol and chk["orth_vh"] <= 10 * tol
                       and chk["triplet_resid_over_s0"] <= 4 * tol and s_all.shape == s_ref.shape)
      rec["check"] = chk
    rows.append(rec)

  g = svd_case_matrix(be, n, "graded", seed=3 + n)            # s_i = 2^(-i/32): the tail norm picks k ~ n / 16
  err = 2.0 ** (-(n // 16) / 32.0) * 4.7
  run("err_only", "graded", g, lambda x: be.svd(x, 1, max_truncation_error=err, relative=True))
  del g
  nf = min(n, 2048)
  a = svd_case_matrix(be, nf, "gauss", seed=3 + nf)
  run("full", "gauss", a, lambda x: be.svd(x, 1))
  del a
  nodd = nf - 48 + 8                                            # 2008 = 16 * 125 + 8: padded to the 16-wide panels
  a = svd_case_matrix(be, nodd, "gauss", seed=3 + nodd)
  run("pad16", "gauss", a, lambda x: be.svd(x, 1, max_singular_values=nodd // 16))
  del a
  a64 = be.cast(svd_case_matrix(be, n, "gauss", seed=3 + n), np.float64)       # the sweep's Gaussian matrix, in f64
  run("max_sv", "gauss", a64, lambda x: be.svd(x, 1, max_singular_values=n // 16), dtype="f64", tol=1e-11, tol_rest=3e-8)
  del a64
  be.lib.tnh_trim()
  return rows


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


# --------------------------------------------------------------------------- sliced network
def sliced_network_bench(ta, be, comm, rank, world, D, min_slices, verify):
  """64-node random 3-regular network, bf16, bond-sliced greedy contraction."""
  from tensornetwork_amd import distributed, workloads  # pylint: disable=import-outside-toplevel
  n = 64
  tensors = workloads.random_regular_device_tensors(be, n, D, ta.bfloat16, seed=6)
  nodes = workloads.random_regular_network(be, n=n, D=D, seed=6, tensors=tensors)
  cuts = distributed.choose_cut_edges(nodes, min_slices=min_slices, world=world)
  rep = distributed.slicing_report(nodes, cuts, world=world)
  cut_at = [[(nodes.index(nd), ax) for nd, ax in e.ends()] for e in cuts]     # the same cuts on a copy of the network

  class _Timed:
    """wraps the communicator: separates this rank's compute time from the all-reduce"""
    def __init__(self, inner):
      self.inner, self.rank, self.world = inner, (inner.rank if inner else 0), (inner.world if inner else 1)
      self.t_compute_end = self.t_reduce_end = None
    def all_reduce_sum(self, backend, tensor):
      backend.synchronize()
      self.t_compute_end = time.perf_counter()
      out = self.inner.all_reduce_sum(backend, tensor) if self.inner is not None else tensor
      backend.synchronize()
      self.t_reduce_end = time.perf_counter()
      return out

  class _Few(distributed.LocalComm):   # warm-up on half of the slices (allocator, kernels)
    rank, world = 0, max(1, int(rep["n_slices"]) // 2)
  kw, staged_error = {}, None
  try:
    distributed.contract_sliced(nodes, cuts, comm=_Few())
  except Exception as exc:  # pylint: disable=broad-except
    if world != 1:
      raise               # (a fallback on one rank only would change the partition under the others)
    # the default mode (every step once per value of the cuts it depends on) failed on this box: slice by slice
    staged_error = f"{type(exc).__name__}: {exc}"[:300]
    kw = {"reuse": False}
    rep["staged_by_default"] = False
    distributed.contract_sliced(nodes, cuts, comm=_Few(), **kw)
  sync_all(be, comm)
  timed = _Timed(comm)
  ran = {}
  t0 = time.perf_counter()
  out = distributed.contract_sliced(nodes, cuts, comm=timed, stats=ran, **kw)
  sync_all(be, comm)
  t = time.perf_counter() - t0
  t_compute = timed.t_compute_end - t0
  t_reduce = timed.t_reduce_end - timed.t_compute_end
  if comm is not None:
    t = comm.max_over_ranks(t)
    t_compute_max = comm.max_over_ranks(t_compute)
    t_reduce_max = comm.max_over_ranks(t_reduce)
  else:
    t_compute_max, t_reduce_max = t_compute, t_reduce
  # The cost model counts multiply-adds.  EXECUTED work: contract_sliced runs every step once per distinct value of the
  # cut bonds it depends on (D = 12: 99.4 % of a slice's work depends on ONE of the two cuts, i.e. 12 runs, not 144);
  # without that mode the slice-invariant steps still run once per rank, not once per slice.
  # WHAT RAN: the mode and the executed multiply-adds come from the timed call's own counters -- every
  # rank's, summed -- not from the host cost model; the model's figure stays beside it as a cross-check.
  # the SAME network and cuts on ONE GPU (every rank runs it, concurrently and without any exchange; rank 0's time is
  # the record): the denominator of the strong-scaling claim, measured in the same process minutes apart
  if world > 1:
    be.synchronize()
    t1 = time.perf_counter()
    alone = distributed.contract_sliced(nodes, cuts, comm=distributed.LocalComm(), **kw)
    be.synchronize()
    t_one = time.perf_counter() - t1
    del alone
    t_one = comm.max_over_ranks(t_one if rank == 0 else 0.0)
  else:
    t_one = t
  alone_flops = 2.0 * rep["flops_per_slice"] * rep["n_slices"]
  executed = float(ran.get("executed_macs", 0.0))
  if comm is not None:
    executed = comm.sum_over_ranks(executed)
  total_flops = 2.0 * executed
  staged = ran.get("mode") == "staged"
  model_flops = 2.0 * rep["flops_with_reuse_all_ranks"] if rep.get("staged_by_default") else None
  result = float(np.asarray(out).reshape(-1)[0])
  rec = {"workload": f"64-node random 3-regular network (seed 6), bond D={D}, bf16, {len(cuts)} cut bonds",
         "n_slices": int(rep["n_slices"]), "n_gpus": world, "seconds": t, "scaling": "strong",
         "mode": "every step once per value of the cuts it depends on" if staged else "slice by slice",
         "flops_total": total_flops, "tflops": total_flops / t / 1e12,
         "flops_total_by_the_host_model": model_flops,
         "executed_equals_model": (abs(total_flops - model_flops) <= 1e-6 * model_flops) if (staged and model_flops) else None,
         "slices_per_rank": rep.get("slices_per_rank"),
         "ideal_speedup_of_this_partition": (distributed.slicing_report(nodes, cuts, world=1)["flops_with_reuse_slowest_rank"]
                                             / rep["flops_with_reuse_slowest_rank"]) if rep.get("flops_with_reuse_slowest_rank") else None,
         "flops_if_every_slice_ran_alone": alone_flops, "speedup_over_slices_alone_at_this_rate": alone_flops / total_flops,
         "peak_intermediate_elems": rep["peak_per_slice"],
         "steps_per_slice": int(rep.get("steps_per_slice", 0)) - int(rep.get("invariant_steps", 0)),
         "slice_invariant_steps_run_once": int(rep.get("invariant_steps", 0)),
         "seconds_1gpu_same_cuts": t_one, "speedup_over_1gpu_same_cuts": t_one / t,
         "slowest_rank_compute_seconds": t_compute_max, "allreduce_seconds": t_reduce_max,
         "collective": ("none" if world == 1 else "one all-reduce(sum) of the fp32-accumulated scalar ON THE HOST (TCP; the RCCL "
                        "communicator did not come up)" if isinstance(comm, HostComm) else
                        "one all-reduce(sum) of the fp32-accumulated scalar (tnh_allreduce, RCCL)"),
         "accumulation": "slice partials added in fp32, rounded to bf16 once", "result": result}
  if staged_error is not None:
    rec["default_mode_error"] = staged_error
  if verify and world == 1:
    # the same slices in f32 on the same (bf16-valued) tensors, slice partial by slice partial, against the a-priori
    # error model of partials_check (the round-2 tolerance of 5e-2 |ref| on the SUM was written
    # after the number was seen)
    t0 = time.perf_counter()
    p16, p32 = [], []
    distributed.contract_sliced(nodes, cuts, partials_out=p16, **kw)
    t32 = [be.cast(x, np.float32) for x in tensors]
    nodes32 = workloads.random_regular_network(be, n=n, D=D, seed=6, tensors=t32)
    cuts32 = [nodes32[ends[0][0]][ends[0][1]] for ends in cut_at]      # the same bonds (no second search)
    ref = float(np.asarray(distributed.contract_sliced(nodes32, cuts32, partials_out=p32, **kw), dtype=np.float64).reshape(-1)[0])
    be.synchronize()
    chk = partials_check(p16, p32, n - 1)             # n nodes: n - 1 intermediates per slice
    chk.update({"f32_same_slices": ref, "bf16": result, "rel_err_of_the_sum": abs(result - ref) / max(abs(ref), 1e-30),
                "note": "the sum of the slice partials cancels (random signs), so its relative error is larger than the "
                        "partials' and is reported, not judged", "f32_seconds": time.perf_counter() - t0})
    rec["verified"] = chk
  return rec


# --------------------------------------------------------------------------- sweeps
def timed_steps(be, fn, reps, batches=3):
  """Mean seconds per call over `reps` back-to-back calls, best of `batches` batches (secondary rows only: a
  batch of ten 0.1 ms contractions is 1 ms long, one host hiccup in it reads as a 6x slower kernel -- seen
  once on the D = 64 row; the headline is timed over exactly K steps, never best-of)."""
  # warm-up for at least 50 ms: the first milliseconds after an idle period (tnh_trim, a host-side check) run at a low
  # shader clock -- round-3 run: the first layout of D = 32 / 64 read 3-4x slow while the second one was normal
  t_w = time.perf_counter()
  while True:
    out = fn()
    del out
    be.synchronize()
    if time.perf_counter() - t_w >= 0.05:
      break
  p0 = be.permute_launches
  best = float("inf")
  for _ in range(batches):
    t0 = time.perf_counter()
    for _ in range(reps):
      out = fn()
      del out
    be.synchronize()
    best = min(best, (time.perf_counter() - t0) / reps)
  return best, (be.permute_launches - p0) / (reps * batches)


def bond_sweep(ta, be, verify):
  """Whole-path TFLOP/s of contract_between over the bond-dimension sweep (rank 0, N = 1 only)."""
  rows, checks = [], {}
  for D in (32, 64, 96, 128, 192, 256):
    A, B = make_nodes(ta, be, D, "L0", seed=7, fill="normal")
    for layout in ("L0", "L1"):
      reps = 3 if D >= 192 else 10
      t, permutes = timed_steps(be, lambda: one_step(ta, be, A, B, layout), reps,   # pylint: disable=cell-var-from-loop
                                batches=1 if D >= 192 else 3)
      rows.append({"D": D, "layout": layout, "gemm": [D * D] * 3, "ms": t * 1e3, "tflops": 2.0 * D**6 / t / 1e12,
                   "kernel": be.lib.tnh_gemm_last_kernel().decode(), "permute_launches": permutes})
      if verify and D in (192, 256) and not (D == 256 and layout == "L0"):   # D=256 L0 is the headline's own check
        out = one_step(ta, be, A, B, layout)
        checks[f"D{D}_{layout}"] = verify_pair(be, A, B, out.tensor, layout, seed=D)
        del out
    del A, B
  # north-star row: two shared D = 512 bonds, M = N = 8192, K = 262144
  A = be.device_random((64, 128, 512, 512), dtype=ta.bfloat16, seed=11, normal=True, a=0.0, b=1.0 / 512)
  B = be.device_random((512, 512, 128, 64), dtype=ta.bfloat16, seed=12, normal=True, a=0.0, b=1.0 / 512)

  def step():
    a, b = ta.Node(A, backend=be), ta.Node(B, backend=be)
    a[2] ^ b[0]  # pylint: disable=pointless-statement
    a[3] ^ b[1]  # pylint: disable=pointless-statement
    return ta.contract_between(a, b)
  t, permutes = timed_steps(be, step, 3, batches=1)
  tf = 2.0 * 8192 * 8192 * 262144 / t / 1e12
  rows.append({"D": 512, "layout": "A(64,128,512,512).B(512,512,128,64)", "gemm": [8192, 8192, 262144], "ms": t * 1e3,
               "tflops": tf, "frac_of_bf16_peak": tf / BF16_MFMA_PEAK_TFLOPS,
               "kernel": be.lib.tnh_gemm_last_kernel().decode(), "permute_launches": permutes})
  if verify:
    out = step()
    checks["D512_row"] = verify_pair(be, A, B, out.tensor, "L0", seed=512)
    del out
  return rows, checks


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
    t, _ = timed_steps(be, lambda: be.tensordot(A, B, [[2, 3], [0, 1]]), 5)   # pylint: disable=cell-var-from-loop
    kernel = be.lib.tnh_gemm_last_kernel().decode()
    row = {"dtype": name, "D": D, "gemm": [D * D] * 3, "ms": t * 1e3, "tflops": mult * D**6 / t / 1e12,
           "kernel": kernel}
    # which pipe did the work, and what fraction of ITS peak is that
    if name == "f64" or name == "complex128":
      row["roofline"] = {"bound": "mfma f64", "peak": F64_MFMA_PEAK_TFLOPS, "frac": row["tflops"] / F64_MFMA_PEAK_TFLOPS}
    elif "bf16" in kernel or "pp" in kernel:
      # f32 (and complex64 through the real expansion) on the bf16 cores: six bf16 products per f32 product
      bf16_tf = 6.0 * row["tflops"]
      row["roofline"] = {"bound": "mfma bf16 (3 x bf16 split: 6 bf16 flop per f32 flop)", "bf16_tflops": bf16_tf,
                         "peak": BF16_MFMA_PEAK_TFLOPS, "frac": bf16_tf / BF16_MFMA_PEAK_TFLOPS,
                         "note": "f32-equivalent TFLOP/s may exceed the 157 TF f32-MFMA peak; the bound is the bf16 pipe"}
    else:
      row["roofline"] = {"bound": "mfma f32", "peak": F32_MFMA_PEAK_TFLOPS, "frac": row["tflops"] / F32_MFMA_PEAK_TFLOPS}
    rows.append(row)
    del A, B
  return rows


# --------------------------------------------------------------------------- configs[3]
def mps_chain_bench(ta, be, cpu):
  """configs[3]: <psi|psi> of a 16-site MPS (32 nodes: kets + conjugates), bulk bond D = 512, f32,
  contractors.greedy -- launch-latency-bound (about 1e9 flop at d = 2), so also the heavier d = 4 variant;
  eager and hipGraph replay.  No exchange step: N GPUs run N independent replicas."""
  from tensornetwork_amd import contractors, workloads as wl  # pylint: disable=import-outside-toplevel
  rows = []
  for d in (2, 4):
    kets = wl.mps_tensors(16, d, 512, seed=5, dtype=np.float32)
    dev = [be.convert_to_tensor(k) for k in kets]
    run = lambda b, ts: contractors.greedy(wl.mps_overlap_network(b, ts)).tensor
    out = run(be, dev)
    be.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
      out = run(be, dev)
    be.synchronize()
    t_eager = (time.perf_counter() - t0) / reps
    g = be.capture(lambda *ts: run(be, list(ts)), *dev)
    try:
      g.launch()
      be.synchronize()
      t0 = time.perf_counter()
      for _ in range(reps):
        o2 = g.launch()
      be.synchronize()
      t_graph = (time.perf_counter() - t0) / reps
      v_graph = float(np.asarray(o2[0] if isinstance(o2, (list, tuple)) else o2))
    finally:
      g.close()
    rec = {"sites": 16, "d": d, "D": 512, "dtype": "f32", "gpu_eager_ms": t_eager * 1e3, "gpu_graph_replay_ms": t_graph * 1e3,
           "value_gpu": float(np.asarray(out)), "value_graph": v_graph}
    if cpu:
      from oracle import numpy_oracle as orc  # pylint: disable=import-outside-toplevel
      ob = orc.OracleBackend()
      run(ob, kets)
      t0 = time.perf_counter()
      ref = float(run(ob, kets))
      rec["cpu_ms"] = (time.perf_counter() - t0) * 1e3
      rec["value_cpu"] = ref
      rec["rel_err_vs_cpu"] = abs(rec["value_gpu"] - ref) / abs(ref)
      rec["ok"] = bool(rec["rel_err_vs_cpu"] <= 1e-4 and abs(v_graph - ref) <= 1e-4 * abs(ref))
    rows.append(rec)
  return rows


# --------------------------------------------------------------------------- configs[4]
def mera_bench(ta, be, chi, verify=False):
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
  p0 = be.permute_launches
  t0 = time.perf_counter()
  out = run()
  be.synchronize()
  t = time.perf_counter() - t0
  rec = {"workload": f"binary-MERA layer energy, chi={chi}, bf16, left + right placement, contractors.branch(nbranch=2)",
         "seconds": t, "flops": 4.0 * float(macs), "tflops": 4.0 * float(macs) / t / 1e12,
         "peak_intermediate_elems": float(peak), "permute_launches": be.permute_launches - p0,
         "energy": float(np.asarray(out).reshape(-1)[0])}
  if verify:
    # the same network in f32 on the GPU.  Its chi^7 intermediates do not fit in f32 (2 x 137 GB),
    # so both runs are bond-sliced (>= chi slices, peak chi^6): the SAME slices in bf16 and in f32 on the same
    # values, compared slice by slice with the a-priori model of partials_check.  The unsliced bf16 energy above
    # (the timed run) is reported next to the sum of the f32 partials.
    from tensornetwork_amd import distributed  # pylint: disable=import-outside-toplevel
    t0 = time.perf_counter()
    parts = {"bf16": [], "f32": []}
    n_round = 0
    for name in ("bf16", "f32"):
      ts = [ham, rho, iso, dis] if name == "bf16" else [be.cast(x, np.float32) for x in (ham, rho, iso, dis)]
      for pl in ("left", "right"):
        nd = workloads.mera_layer_network(be, *ts, pl)
        cuts = distributed.choose_cut_edges(nd, min_slices=chi)
        distributed.contract_sliced(nd, cuts, partials_out=parts[name])
        n_round = len(nd) - 1
        for x in nd:
          x.tensor = None
      del ts
    chk = partials_check(parts["bf16"], parts["f32"], n_round)
    e32 = 0.5 * float(np.sum(np.concatenate([p.reshape(-1) for p in parts["f32"]])))
    chk.update({"bf16_unsliced_energy": rec["energy"], "f32_sliced_energy": e32,
                "rel_diff_of_the_scalars": abs(rec["energy"] - e32) / max(abs(e32), 1e-30),
                "f32_seconds": time.perf_counter() - t0})
    rec["verified"] = chk
  return rec


def verify_mera(ta, be, chi=16):
  """The MERA layer energy at chi = 16 from the SAME device-generated tensors three ways: float64 on the
  oracle backend (host), f32 on the GPU, bf16 on the GPU."""
  from oracle import numpy_oracle as orc  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import contractors, workloads  # pylint: disable=import-outside-toplevel
  sc = lambda n: float(n) ** -0.5
  shapes = [((chi,) * 6, sc(chi**3)), ((chi,) * 6, sc(chi**3)), ((chi,) * 3, sc(chi)), ((chi,) * 4, sc(chi * chi))]
  dev16 = [be.device_random(s, dtype=ta.bfloat16, seed=31 + i, normal=True, b=b) for i, (s, b) in enumerate(shapes)]
  contractor = lambda nd: contractors.branch(nd, nbranch=2)
  e16 = float(np.asarray(workloads.mera_energy(be, *dev16, contractor)).reshape(-1)[0])
  dev32 = [be.cast(t, np.float32) for t in dev16]
  e32 = float(np.asarray(workloads.mera_energy(be, *dev32, contractor)).reshape(-1)[0])
  host = [host64(t) for t in dev16]
  t0 = time.perf_counter()
  ref = float(workloads.mera_energy(orc.OracleBackend(), *host, contractor))
  t_cpu = time.perf_counter() - t0
  # scale: the energy is a sum of products with cancellations; judge the error against the rms of the
  # two placement terms' magnitude proxy |ref| + tiny
  rel32 = abs(e32 - ref) / max(abs(ref), 1e-30)
  rel16 = abs(e16 - ref) / max(abs(ref), 1e-30)
  return {"chi": chi, "oracle_f64": ref, "hip_f32": e32, "hip_bf16": e16, "rel_err_f32": rel32, "rel_err_bf16": rel16,
          "oracle_seconds": t_cpu, "tol": "f32 1e-3, bf16 (11 bf16-rounded intermediates) 1e-1 relative",
          "ok": bool(rel32 <= 1e-3 and rel16 <= 1e-1)}


def mera_chi64_bench(ta, be, n_gpus=8, verify=False, full_placements=0, budget_s=200.0, chi=64):
  from tensornetwork_amd import workloads  # pylint: disable=import-outside-toplevel
  per = workloads.mera_sliced_sample(be, chi, ta.bfloat16, reps=2)
  measured = {}
  for pl in ("left", "right")[:max(0, int(full_placements))]:
    # the placement as a RUN: all chi^2 slices through distributed._contract_slices_staged (the machinery of
    # contract_sliced), partials added on the device.  Every step runs once per distinct value of the cuts it depends
    # on, with BOTH ends of each cut leg sliced (slice_edge): at chi = 64 the class that depends on both cuts is 98 %
    # of the executed work, and the executed multiply-adds equal the flop-optimal dense cost of the network.
    try:
      run = workloads.mera_sliced_run(be, chi, pl, ta.bfloat16, budget_seconds=budget_s, check_every=1024 if verify else 0)
    except Exception as exc:  # pylint: disable=broad-except
      run = workloads.mera_sliced_run(be, chi, pl, ta.bfloat16, budget_seconds=budget_s, check_every=1024 if verify else 0,
                                      reuse_partials=False)
      run["reuse_partials_error"] = f"{type(exc).__name__}: {exc}"[:300]
    if verify and run["checks"]:
      chk = partials_check([[c[1]] for c in run["checks"]], [[c[2]] for c in run["checks"]], 11)
      run["checks_vs_f32"] = {k: chk[k] for k in ("n_values", "rms_rel_err", "model_rms", "err_over_tol", "ok")}
    measured[pl] = run
  checked = None
  if verify:
    # four slices per placement of the layer workloads.MeraSlicedLayer defines (slice_edge semantics: both ends of each
    # cut leg sliced), each contracted alone, bf16 vs f32 on the same values, a-priori rounding bound per slice
    vals = workloads.mera_slice_values(be, chi, [(0, 0), (1, 5 % chi), (17 % chi, 3 % chi), (chi - 1, chi - 2)], ta.bfloat16)
    rows, p16, p32 = [], [], []
    for pl, v in vals.items():
      for r in v["rows"]:
        rows.append({"placement": pl, "slice": r["slice"], "bf16": r["half"], "f32": r["f32"]})
        p16.append([r["half"]])
        p32.append([r["f32"]])
    checked = partials_check(p16, p32, 11)          # 12 nodes: 11 intermediates
    checked["slices"] = rows
  total = sum(v["sec_per_slice"] * v["n_slices"] for v in per.values())
  flops = sum(2.0 * v["macs_per_slice"] * v["n_slices"] for v in per.values())
  rec = {"workload": f"binary-MERA layer energy at chi = {chi}, bf16: two cut bonds per placement -> {chi * chi} slices each "
                     "(the dense network needs 137 GB rank-6 inputs and chi^7 intermediates)",
         "placements": per, "flops_total": flops,
         "layer_seconds_1gpu_extrapolated": total, f"layer_seconds_{n_gpus}gpu_extrapolated": total / n_gpus,
         "tflops_1gpu": flops / total / 1e12,
         "label": "EXTRAPOLATED rows: measured seconds per slice (one slice per placement, best of 3) x slice count; "
                  "slices are independent, one scalar all-reduce at the end"}
  if measured:
    done = sum(m["slices_done"] for m in measured.values())
    secs = sum(m["seconds"] for m in measured.values())
    rec["measured"] = measured
    rec["measured_slices"] = done
    rec["measured_seconds"] = secs
    # executed multiply-adds (with partial results reused across slices far fewer than slices x one slice's cost)
    rec["measured_tflops"] = sum(2.0 * m.get("executed_macs", m["macs_per_slice"] * m["slices_done"])
                                 for m in measured.values()) / secs / 1e12
    rec["measured_reuse_partials"] = bool(all(m.get("reuse_partials") for m in measured.values()))
    alone = sum(per[pl]["sec_per_slice"] * m["slices_done"] for pl, m in measured.items() if pl in per)
    if alone > 0:
      rec["measured_speedup_over_slice_by_slice"] = alone / secs       # against (seconds of one slice alone) x slices done
    complete = all(m["slices_done"] == m["n_slices"] for m in measured.values())
    rec["measured_executed_macs"] = sum(m.get("executed_macs", 0.0) for m in measured.values())
    rec["measured_macs_if_every_slice_ran_alone"] = sum(m["macs_per_slice"] * m["slices_done"] for m in measured.values())
    rec["measured_label"] = (f"MEASURED on 1 GPU: {len(measured)} of 2 placements, every slice contracted "
                             f"({done} slices{'' if complete else ', stopped by the time budget'}) through "
                             "distributed._contract_slices_staged; each cut slices BOTH tensors on its leg (slice_edge), "
                             "a step runs once per distinct value of the cuts it depends on when "
                             "measured_reuse_partials is true -- a complete placement then executes the flop-optimal "
                             "DENSE cost of the network (3.72e16 multiply-adds at chi = 64), 41 % of 4096 stand-alone "
                             "slices (the EXTRAPOLATED rows are per-slice seconds x slices, without reuse); the 8-GPU "
                             "figure stays arithmetic (no 8-GPU node on the builder's side)")
    if verify:
      rec.setdefault("verified_runs", {pl: m.get("checks_vs_f32") for pl, m in measured.items()})
    # the same placement on n_gpus ranks: host arithmetic on the plan (which slices each rank gets, what it executes) --
    # a MODEL row; the one-GPU rehearsal of all 8 shares is profiles/r05_mera_rank_share_rehearsal.jsonl (7.63x of 7.71x)
    try:
      import itertools  # pylint: disable=import-outside-toplevel
      pl0 = next(iter(measured))
      stage = workloads._mera_slice_plan(chi, pl0)["stage"]      # pylint: disable=protected-access
      every = list(itertools.product(range(chi), range(chi)))
      blocks = stage.partition(every, n_gpus)
      ideal = stage.macs_with_reuse(every) / max(stage.macs_with_reuse(b) for b in blocks)
      if measured[pl0]["slices_done"] == measured[pl0]["n_slices"]:
        rec["ranks_model"] = {"world": n_gpus, "ideal_speedup_of_the_partition": ideal,
                              "cut_values_per_rank": [len({i for i, _ in blocks[0]}), len({j for _, j in blocks[0]})],
                              "seconds_per_placement_at_ideal": measured[pl0]["seconds"] / ideal,
                              "label": "MODEL (slowest rank's executed multiply-adds); rehearsed on one GPU by "
                                       "tools/mera_rank_share_rehearsal.py"}
    except Exception as exc:  # pylint: disable=broad-except
      rec["ranks_model_error"] = f"{type(exc).__name__}: {exc}"[:200]
  if checked is not None:
    rec["verified"] = checked
  return rec


# --------------------------------------------------------------------------- helper kernels
def helpers_bench(ta, be):
  """HBM-bound helper kernels at the shapes the BASELINE configs produce; GB/s = algorithmic bytes / time.

  Every row ROTATES over enough distinct input buffers, and keeps as many results alive, that the bytes touched
  between two uses of the same buffer exceed 1 GiB = 4x the 256 MiB Infinity Cache (the round-3
  rows re-ran one 67 MB input with a recycled output block -- those were Infinity-Cache rates, not HBM rates)."""
  from tensornetwork_amd import _lib  # pylint: disable=import-outside-toplevel
  rows = []

  def add(name, make, op, nbytes, touched, reps=3):
    n_buf = max(2, -(-(1 << 30) // touched))
    xs = [make(i) for i in range(n_buf)]
    outs = [op(x) for x in xs]            # warm-up; the results stay alive: every call of a rotation writes its own block
    be.synchronize()
    s = _lib.Event().record()
    for _ in range(reps):
      for i, x in enumerate(xs):
        outs[i] = None                    # back to the pool just before the call that takes a block of this size again
        outs[i] = op(x)
    e = _lib.Event().record()
    e.synchronize()
    ms = s.elapsed_ms(e) / (reps * n_buf)
    rows.append({"op": name, "ms": ms, "algorithmic_bytes": nbytes, "gbps": nbytes / ms / 1e6,
                 "hbm_roofline_frac": nbytes / ms / 1e6 / HBM_PEAK_GBPS, "buffers": n_buf,
                 "bytes_between_reuse": n_buf * touched})
    del xs, outs
    _lib.check(be.lib.tnh_trim())

  f32_16_6 = 4 * 16 ** 6
  add("K1 permute f32 (16,)*6 -> (0,2,4,1,3,5)  [configs[2] mixed edge order]",
      lambda i: be.device_random((16,) * 6, dtype=np.float32, seed=41 + i),
      lambda x: be.transpose(x, (0, 2, 4, 1, 3, 5)), 2 * f32_16_6, 2 * f32_16_6)
  bf16_128_4 = 2 * 128 ** 4
  mk = lambda i: be.device_random((128,) * 4, dtype=ta.bfloat16, seed=52 + i)      # noqa: E731
  add("K1 permute bf16 (128,)*4 -> (0,2,1,3)  [configs[1] L1 operand, D = 128]", mk,
      lambda x: be.transpose(x, (0, 2, 1, 3)), 2 * bf16_128_4, 2 * bf16_128_4)
  add("K1 permute bf16 (128,)*4 -> (2,3,0,1)  [[K][N] -> [N][K]]", mk,
      lambda x: be.transpose(x, (2, 3, 0, 1)), 2 * bf16_128_4, 2 * bf16_128_4)
  big = 4 * 4096 * 4096 * 16
  mk = lambda i: be.device_random((4096, 4096, 16), dtype=np.float32, seed=63 + i)  # noqa: E731
  add("K4 sum f32 (4096,4096,16) over axis 1", mk, lambda x: be.sum(x, axis=1), big + big // 4096, big)
  add("K4 sum f32 (4096,4096,16) over all axes", mk, lambda x: be.sum(x), big, big)
  tr = 4 * 4096 * 256 * 256
  add("K3 trace f32 (4096,256,256) over the last two axes",
      lambda i: be.device_random((4096, 256, 256), dtype=np.float32, seed=74 + i), lambda x: be.trace(x),
      4096 * 256 * 4 + 4096 * 4, 4096 * 256 * 128)
  # a diagonal element sits alone in its 128-byte line (stride 257 floats): the memory system moves a line per element
  rows[-1]["line_granular_bytes"] = 4096 * 256 * 128 + 4096 * 4
  rows[-1]["line_granular_gbps"] = rows[-1]["line_granular_bytes"] / rows[-1]["ms"] / 1e6
  rows[-1]["note"] = "gbps counts the 4 useful bytes per diagonal element; line_granular_gbps the 128-byte lines they arrive in"
  sq = 4 * 4096 * 4096
  mk = lambda i: be.device_random((4096, 4096), dtype=np.float32, seed=85 + i)     # noqa: E731
  v = be.device_random((4096,), dtype=np.float32, seed=46)
  add("K5 broadcast_right_multiplication f32 4096^2 x (4096,)  [u sqrt(s)]", mk,
      lambda x: be.broadcast_right_multiplication(x, v), 2 * sq, 2 * sq)
  add("K5 broadcast_left_multiplication f32 (4096,) x 4096^2  [sqrt(s) vh]", mk,
      lambda x: be.broadcast_left_multiplication(v, x), 2 * sq, 2 * sq)
  add("K6 sqrt f32 4096^2", mk, lambda x: be.sqrt(x), 2 * sq, 2 * sq)
  return rows


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


GATHER_CASES = {     # contracted axes of the rank-8 long tensor (bond D): where they sit decides what a box looks like
    "k37_innermost_contracted_free_run_1728": [3, 7],
    "k47_innermost_contracted_free_run_144": [4, 7],
    "k16_innermost_free_contracted_right_above": [1, 6],
    "k15_two_innermost_free": [1, 5],
    "k01_k_major": [0, 1],
    # 1728 contracted indices (the 144 x 248 832 x 1728 product of a slice): the K-loop kernel, 12 steps of 144
    "k167_k_loop_1728": ([1, 3, 4], [1, 6, 7]),
}


def gather_gemm_bench(ta, be, verify, D=12, rank=8, reps=5, cases=None):
  """A D x D x D x D tensor takes two bonds off a rank-8 intermediate (D = 12: 430 M elements, the 144 x 2 985 984
  x 144 products of the north-star network): `tnh_gemm_gather` reading the intermediate where it lies against the
  classic lowering (K1 permute + streaming GEMM), both operand orders, whole-call microseconds (host side included).
  Results are compared on the device: same MFMA sequence per element, so the difference must be exactly zero."""
  from tensornetwork_amd import hip_backend  # pylint: disable=import-outside-toplevel
  long_shape = (D,) * rank
  small4 = be.device_random((D, D, D, D), dtype=ta.bfloat16, seed=1, normal=True, a=0.0, b=0.1)
  small5 = None                                             # (made when a case contracts three axes)
  long_ = be.device_random(long_shape, dtype=ta.bfloat16, seed=2, normal=True, a=0.0, b=0.1)
  keep = (be.gather_gemm, be.gather_min_rows)
  be.gather_min_rows = min(be.gather_min_rows, D**(rank - 3))

  def timed(fn):
    out = fn()
    del out
    be.synchronize()
    best = float("inf")
    for _ in range(3):
      t0 = time.perf_counter()
      for _ in range(reps):
        out = fn()
        del out
      be.synchronize()
      best = min(best, (time.perf_counter() - t0) / reps)
    return best

  rows, all_exact = [], True
  try:
    for name, spec in (cases or GATHER_CASES).items():
      axes_s, axes_l = spec if isinstance(spec, tuple) else ([1, 3], spec)
      if len(axes_s) == 3 and small5 is None:
        small5 = be.device_random((D,) * 5, dtype=ta.bfloat16, seed=3, normal=True, a=0.0, b=0.1)
      small = small5 if len(axes_s) == 3 else small4
      # read the long operand and the small one, write the result
      nbytes = 2 * (D**rank + D * D * D**(rank - len(axes_l)) + small.size)
      plan = hip_backend._gather_descriptor(long_shape, axes_l)      # pylint: disable=protected-access
      row = {"case": name, "contracted_axes": axes_l, "algorithmic_bytes": nbytes,
             "box": None if plan is None else {"rows": plan[1], "piece_bytes": hip_backend._gather_piece_bytes(plan[0]),  # pylint: disable=protected-access
                                               "innermost_axis_contracted": bool(plan[0].k_mask & 1)}}
      for orient in ("small_first", "long_first"):
        call = (small, long_, [axes_s, axes_l]) if orient == "small_first" else (long_, small, [axes_l, axes_s])
        rec = {}
        for mode in ("classic", "gather"):
          be.gather_gemm = mode == "gather"
          g0, p0 = be.gather_launches, be.permute_launches
          t = timed(lambda: be.tensordot(*call))      # pylint: disable=cell-var-from-loop
          n_calls = 3 * reps + 1
          rec[mode + "_us"] = t * 1e6
          rec[mode + "_TBps"] = nbytes / t / 1e12
          rec[mode + "_launches"] = {"gather": (be.gather_launches - g0) // n_calls, "permute": (be.permute_launches - p0) // n_calls}
          rec[mode + "_kernel"] = be.lib.tnh_gemm_last_kernel().decode()
        rec["speedup"] = rec["classic_us"] / rec["gather_us"]
        if verify:
          be.gather_gemm = True
          got = be.tensordot(*call)
          be.gather_gemm = False
          ref = be.tensordot(*call)
          rec["max_abs_difference"] = float(np.asarray(be.norm(be.subtraction(got, ref))).reshape(-1)[0])
          looped = plan is not None and plan[0].kl_ext > 1
          if looped:      # another grouping of the fp32 partial sums than the tile kernels': equal to round-off, not bit for bit
            rec["rel_difference"] = rec["max_abs_difference"] / max(float(np.asarray(be.norm(ref)).reshape(-1)[0]), 1e-30)
          del got, ref
          trailing = sorted(axes_l) == list(range(rank - len(axes_l), rank))     # the streaming kernel's own case
          same = rec["rel_difference"] <= 2.0**-9 if looped else rec["max_abs_difference"] == 0.0
          all_exact = all_exact and same and rec["gather_launches"]["gather"] == (0 if trailing or plan is None else 1)
        row[orient] = rec
      rows.append(row)
  finally:
    be.gather_gemm, be.gather_min_rows = keep
  out = {"workload": f"bf16 ({D},)*4 x ({D},)*{rank}, two bonds contracted (k_loop row: ({D},)*5, three), rows = contracted "
                     "axes of the long tensor",
         "timing": f"whole tensordot call, mean of {reps}, best of 3", "rows": rows}
  if verify:
    out["verified"] = {"tol": "gather result - classic result == 0 (2-norm of the difference on the device; K loop: relative "
                              "2-norm <= 2^-9), one gather launch", "ok": bool(all_exact)}
  return out


def partials_check(p_half, p_f32, n_roundings, half_bits=9, margin=3.0):
  """bf16 (half_bits = 9: unit round-off u = 2^-9) slice partials against the same slices in f32, judged by an
  A-PRIORI error model instead of a tolerance picked after the fact:
  every intermediate of the path is rounded once (relative error uniform in +-u: variance u^2 / 3); in a contraction
  of random-sign data the relative errors of the operands and of the result's own rounding add in quadrature, so a
  result that depends on n intermediates carries a relative rms error of u sqrt(n / 3).  The statistic is taken over
  the VECTOR of slice partials (a single scalar can be small by cancellation, which says nothing about the
  contraction): rms_rel = |p_half - p_f32|_2 / |p_f32|_2 <= margin * u * sqrt(n / 3).  (The worst-case bound
  ((1 + u)^depth - 1) * value(|tensors|) is rigorous but vacuous here: the network of absolute values is 1e16 times
  the signed one.)"""
  a = np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1) for x in p_half])
  b = np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1) for x in p_f32])
  u = 2.0 ** -half_bits
  model = u * np.sqrt(max(int(n_roundings), 1) / 3.0)
  rms_rel = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
  return {"n_values": int(a.size), "rms_rel_err": rms_rel, "model_rms": float(model), "margin": margin,
          "err_over_tol": rms_rel / (margin * model), "n_roundings": int(n_roundings),
          "tol": "a priori: rms relative error of the slice partials <= margin * 2^-9 * sqrt(n_intermediates / 3)",
          "ok": bool(rms_rel <= margin * model)}


COMPACT_LINE_LIMIT = 4096     # the driver's parser reads the LAST stdout line out of an 8 KB tail


def _num(x, digits=4):
  """a float rounded to `digits` significant digits (the detail file keeps full precision)"""
  if isinstance(x, bool) or not isinstance(x, (int, float)):
    return x
  if isinstance(x, int) or x != x or x in (float("inf"), float("-inf")):
    return x
  return float(f"{x:.{digits}g}")


def _pick(d, keys, digits=4):
  if not isinstance(d, dict):
    return None
  if "error" in d:
    return {"error": str(d["error"])[:120]}
  return {k: _num(d[k], digits) for k in keys if k in d and d[k] is not None}


def compact_line(result, detail_name):
  """The ONE line the driver parses: every contract key, `roofline`, `cpu_baseline`, and a few numbers of each
  secondary leg -- everything else lives in the detail file.  Kept under COMPACT_LINE_LIMIT bytes
  (tests/test_host_lowering_cpu.py checks the length on the emulated backend)."""
  line = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                     "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
  line["value"], line["ms_per_step"] = _num(line["value"], 6), _num(line["ms_per_step"], 6)
  cfg = dict(result.get("config") or {})
  cfg["comm"] = str(cfg.get("comm", "none"))[:48]
  line["config"] = cfg
  roof = result.get("roofline") or {}
  line["roofline"] = {k: _num(roof.get(k), 5) for k in (
      "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes", "kernel",
      "kernel_ms", "launches", "observed_clock_mhz", "board_power_w", "frac_at_observed_clock") if k in roof}
  cpu = result.get("cpu_baseline")
  if isinstance(cpu, dict):
    line["cpu_baseline"] = {k: _num(cpu.get(k)) for k in ("value", "unit", "cores", "kind") if k in cpu}
    if "error" in cpu:
      line["cpu_baseline"]["error"] = str(cpu["error"])[:120]
    if "sample" in cpu:
      line["cpu_baseline"]["sample"] = str(cpu["sample"])[:160]
  ver = result.get("verified")
  if isinstance(ver, dict):
    checks = {k: v for k, v in ver.items() if isinstance(v, dict)}
    line["verified"] = {"all_ok": ver.get("all_ok"), "checks": len(checks),
                        "failed": sorted(k for k, v in checks.items() if not v.get("ok", False))[:8]}
  sweep = result.get("bond_sweep")
  if isinstance(sweep, list):
    # "D<bond><layout>": [TFLOP/s, K1 permute launches per contraction]
    line["bond_sweep"] = {
        f"D{r.get('D')}" + (str(r.get("layout"))[:2] if str(r.get("layout", ""))[:2] in ("L0", "L1") else "row"):
            [_num(r.get("tflops")), r.get("permute_launches")] for r in sweep if isinstance(r, dict)}
  elif isinstance(sweep, dict):
    line["bond_sweep"] = _pick(sweep, ())
  svd = result.get("svd")
  if isinstance(svd, dict):
    line["svd"] = _pick(svd, ("n", "k", "seconds", "gbps", "hbm_roofline_frac", "path"))
    if isinstance(line["svd"], dict) and "error" not in line["svd"]:
      rows = {}
      for r in svd.get("sweep") or []:
        if isinstance(r, dict) and (r.get("mode") or r.get("input") in (None, "gauss")) and r.get("order") in (None, "natural"):
          rows[f"{r.get('dtype', 'f32')}_{r.get('n')}" + (f"_{r['mode']}" if r.get("mode") else "")] = \
              [_num(r.get("seconds")), "b" if str(r.get("path", "")).startswith("band") else "j"]
      line["svd"]["seconds_by_case"] = rows
      if isinstance(svd.get("bound"), dict):
        line["svd"]["bound"] = _pick(svd["bound"], ("launches", "launch_floor_s", "update_floor_s", "sturm_floor_s", "floor_s"))
      if isinstance(svd.get("cpu_baseline"), dict):
        line["svd"]["cpu_gbps"] = _num(svd["cpu_baseline"].get("value"))
  line["rccl_ranks"] = result.get("rccl_ranks", 0)
  sn = result.get("sliced_network")
  if isinstance(sn, dict) and "seconds" in sn:
    # the >= 6x claim of north_star, readable without the detail file: the sliced network on N GPUs against the same
    # network with the same cuts on one of them (`value` above stays the weak-scaling replica aggregate)
    line["strong_scaling"] = {"workload": str(sn.get("workload", ""))[:72], "n_gpus": sn.get("n_gpus"),
                              "seconds": _num(sn.get("seconds")), "seconds_1gpu_same_cuts": _num(sn.get("seconds_1gpu_same_cuts")),
                              "speedup": _num(sn.get("speedup_over_1gpu_same_cuts")),
                              "ideal": _num(sn.get("ideal_speedup_of_this_partition")),
                              "collective": "rccl" if result.get("rccl_ranks", 0) == sn.get("n_gpus") and (sn.get("n_gpus") or 1) > 1
                                            else ("none" if (sn.get("n_gpus") or 1) == 1 else "HOST")}
  line["sliced_network"] = _pick(result.get("sliced_network"),
                                 ("n_slices", "n_gpus", "seconds", "tflops", "mode", "allreduce_seconds", "scaling", "collective",
                                  "permute_time_frac", "ideal_speedup_of_this_partition", "executed_equals_model"))
  line["sliced_network_small"] = _pick(result.get("sliced_network_small"), ("n_slices", "seconds", "tflops", "mode"))
  line["mera"] = _pick(result.get("mera"), ("chi", "seconds", "tflops", "permute_launches"))
  line["mera_chi64"] = _pick(result.get("mera_chi64"),
                             ("measured_seconds", "measured_slices", "measured_tflops", "measured_reuse_partials",
                              "measured_executed_macs", "measured_speedup_over_slice_by_slice", "layer_seconds_1gpu_extrapolated",
                              "layer_seconds_8gpu_extrapolated", "tflops_1gpu"))
  chain = result.get("mps_chain")
  if isinstance(chain, list):
    line["mps_chain_ms"] = {f"d{r.get('d')}": [_num(r.get("gpu_eager_ms")), _num(r.get("gpu_graph_replay_ms"))]
                            for r in chain if isinstance(r, dict)}
  gather = result.get("gather_gemm")
  if isinstance(gather, dict) and isinstance(gather.get("rows"), list):
    # case -> [classic us, gather us] with the small operand first, then with the long operand first
    line["gather_gemm_us"] = {str(r.get("case", "?")).split("_")[0]:
                              [_num(r.get(o, {}).get(m + "_us"), 3) for o in ("small_first", "long_first") for m in ("classic", "gather")]
                              for r in gather["rows"] if isinstance(r, dict)}
  line = {k: v for k, v in line.items() if v is not None or k == "vs_baseline"}
  line["detail"] = detail_name
  text = json.dumps(line, separators=(",", ":"))
  for drop in ("gather_gemm_us", "mps_chain_ms", "sliced_network_small", "mera_chi64", "bond_sweep", "mera", "sliced_network", "svd",
               "strong_scaling"):
    if len(text) < COMPACT_LINE_LIMIT:
      break
    line.pop(drop, None)              # never reached at the bench's own sizes; the contract keys always fit
    text = json.dumps(line, separators=(",", ":"))
  return text


def emit(result, args):
  """Write the full record to the detail file(s) and print the compact line LAST on stdout."""
  detail_name = "bench_detail.json" if args.gpus == 1 else f"bench_detail_n{args.gpus}.json"
  targets = [os.path.join(ROOT, detail_name)]
  scratch = os.path.join(ROOT, "gpurun_out")
  try:
    os.makedirs(scratch, exist_ok=True)
    targets.append(os.path.join(scratch, detail_name))      # travels back from the GPU box with the call
  except OSError:
    pass
  blob = json.dumps(result)
  for path in targets:
    try:
      with open(path, "w") as f:
        f.write(blob + "\n")
    except OSError as exc:
      print(f"[bench] could not write {path}: {exc}", file=sys.stderr)
  print(compact_line(result, detail_name), flush=True)


def fenced(result, key, fn):
  """every secondary leg is fenced: whatever happens in one of them, the headline line is printed"""
  try:
    result[key] = fn()
  except Exception as exc:  # pylint: disable=broad-except
    result[key] = {"error": f"{type(exc).__name__}: {exc}"}


def main():
  args = parse_args()
  if args.gpus < 1:
    sys.exit("--gpus must be >= 1")
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(self_launch(args))       # no launcher around us: start the N ranks ourselves
  rank = int(os.environ.get("RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus and not os.environ.get("TNH_BENCH_FORCE_DIST"):
    # a launcher that started a different number of ranks than --gpus says: never report under the wrong n_gpus
    sys.exit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
  if args.dry_run:
    dry_run(args, rank, world)
    return
  use_dist = world > 1 or bool(os.environ.get("TNH_BENCH_FORCE_DIST"))
  if use_dist:
    # before any HIP runtime comes up in this process: the host driver only supports dmabuf IPC
    # (RCCL / cross-process device memory fail with hipIpcGetMemHandle otherwise); already exported on the boxes
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  if not use_dist or os.environ.get("TNH_BENCH_FORCE_DIST") or world == 1:
    pass
  elif visible_gpus() <= local:
    sys.exit(f"[bench] rank {rank}: LOCAL_RANK {local} has no device (visible: {visible_gpus()})")
  comm, comm_name = None, "none"
  os.environ.setdefault("TNHIP_DEVICE", str(local))
  import tensornetwork_amd as ta  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import _lib, telemetry  # pylint: disable=import-outside-toplevel

  # Collector policy: the package DEFAULT (no gc.freeze(); VERDICT r5 weak 7 -- rounds 4-5 opted in to the frozen
  # baseline here).  Measured on one box, same command with TNH_GC_FREEZE=0 / 1 (round 6): the sweep rows agree within
  # the box's clock noise (D = 96: 1372 / 1366 TFLOP/s, D = 64: 1356 / 1376) now that the allocator amortises its
  # collector passes (device_tensor._grant_slack), so the numbers below are what a user of tn.contract_between gets.
  from tensornetwork_amd import contractors, distributed, pathfinder, workloads  # pylint: disable=import-outside-toplevel,unused-import
  be = ta.get_hip_backend()
  be.lib  # pylint: disable=pointless-statement
  if use_dist:
    from tensornetwork_amd import comm as tcomm  # pylint: disable=import-outside-toplevel
    # the library's own bound on ncclCommInitRank (default 600 s) must fire before the watchdog below does, and a
    # bench run has a driver-side clock around it: 420 s (a cold librccl.so alone was measured at minutes)
    os.environ.setdefault("TNH_COMM_INIT_TIMEOUT_S", "420")
    with Watchdog(args.bringup_timeout, "communicator bring-up (rendezvous, ncclCommInitRank, first barrier)", rank):
      comm, comm_name = bring_up_comm(tcomm, be, rank, world, args.allow_host_exchange)
      sync_all(be, comm)

  D = args.bond
  M = N = K = D * D
  flops_per_step = 2.0 * M * N * K
  A, B = make_nodes(ta, be, D, args.layout, seed=rank, fill=args.fill)

  for _ in range(args.warmup):
    out = one_step(ta, be, A, B, args.layout)
    del out
  tel = telemetry.Telemetry(be.lib) if rank == 0 else None
  be.gemm_events = []          # HIP events around every GEMM launch in the timed region
  p0 = be.permute_launches
  sync_all(be, comm)
  with telemetry.Sampler(tel) as sampler:
    t0 = time.perf_counter()
    for _ in range(args.steps):
      out = one_step(ta, be, A, B, args.layout)
      if _ < args.steps - 1:
        del out
    sync_all(be, comm)
    elapsed = time.perf_counter() - t0
  events, be.gemm_events = be.gemm_events, None
  permutes_per_step = (be.permute_launches - p0) / max(args.steps, 1)
  kernel_name = be.lib.tnh_gemm_last_kernel().decode()
  if comm is not None:
    elapsed = comm.max_over_ranks(elapsed)

  gemm_ms = [s.elapsed_ms(e) for s, e in events]
  gemm_avg_s = (sum(gemm_ms) / len(gemm_ms) / 1e3) if gemm_ms else float("nan")
  achieved = flops_per_step / gemm_avg_s / 1e12 if gemm_ms else float("nan")
  power = sampler.summary() if rank == 0 else {}
  clock = power.get("sclk_mean_mhz")

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
                                                           "(no data-path collective)",
                 "comm": comm_name, "permute_launches_per_step": permutes_per_step,
                 "gc_policy": "package default" if not ta.configure_gc().get("freeze") else "gc.freeze() (TNH_GC_FREEZE / configure_gc)"},
      "roofline": {"bound": "mfma", "achieved": achieved, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": achieved / BF16_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                   "algorithmic_bytes": 2.0 * (M * K + N * K + M * N), "kernel": kernel_name,
                   "kernel_ms": gemm_avg_s * 1e3, "launches": len(gemm_ms),
                   "observed_clock_mhz": clock, "board_power_w": power.get("power_mean_w"),
                   "board_power_cap_w": power.get("power_cap_w"),
                   "peak_at_observed_clock": (BF16_MFMA_PEAK_TFLOPS * clock / MAX_CLOCK_MHZ) if clock else None,
                   "frac_at_observed_clock": (achieved / (BF16_MFMA_PEAK_TFLOPS * clock / MAX_CLOCK_MHZ)) if clock else None},
  }
  result["rccl_ranks"] = rccl_ranks(be, comm)
  verified = {}
  if rank == 0 and not args.no_verify and args.fill == "normal":
    try:
      verified["headline_D%d_%s" % (D, args.layout)] = verify_pair(be, A, B, out.tensor, args.layout, seed=1)
    except Exception as exc:  # pylint: disable=broad-except
      verified["headline"] = {"error": f"{type(exc).__name__}: {exc}"}
  del out, A, B
  ta.trim_pool()
  if args.rr_bond > 0:
    fenced(result, "sliced_network", lambda: sliced_network_bench(ta, be, comm, rank, world, args.rr_bond, args.rr_min_slices,
                                                                  not args.no_verify))
    if isinstance(result["sliced_network"], dict) and "verified" in result["sliced_network"]:
      verified["sliced_network_bf16_vs_f32"] = result["sliced_network"].pop("verified")
    if world == 1 and args.rr_bond_small > 0 and args.rr_bond_small != args.rr_bond:
      # the rounds 1-4 instance (D = 12: 0.092 s on one GPU in round 4), for continuity
      fenced(result, "sliced_network_small", lambda: sliced_network_bench(ta, be, comm, rank, world, args.rr_bond_small,
                                                                          args.rr_min_slices, False))
  if rank == 0:
    single = world == 1
    if single and not args.no_sweep:
      ta.trim_pool()
      try:
        result["bond_sweep"], checks = bond_sweep(ta, be, not args.no_verify)
        verified.update(checks)
      except Exception as exc:  # pylint: disable=broad-except
        result["bond_sweep"] = {"error": f"{type(exc).__name__}: {exc}"}
      ta.trim_pool()
      fenced(result, "dtype_sweep", lambda: dtype_sweep(ta, be))
      ta.trim_pool()
    if single and args.mera_chi > 0:
      fenced(result, "mera", lambda: mera_bench(ta, be, args.mera_chi, verify=not args.no_verify))
      if isinstance(result["mera"], dict) and "verified" in result["mera"]:
        verified[f"mera_chi{args.mera_chi}_bf16_vs_f32"] = result["mera"].pop("verified")
      ta.trim_pool()
    if single and not args.no_extras:
      fenced(result, "mera_chi64", lambda: mera_chi64_bench(ta, be, verify=not args.no_verify,
                                                            full_placements=args.mera64_full, budget_s=args.mera64_budget))
      if isinstance(result["mera_chi64"], dict) and "verified" in result["mera_chi64"]:
        verified["mera_chi64_real_slices_bf16_vs_f32"] = result["mera_chi64"].pop("verified")
      ta.trim_pool()
      fenced(result, "mps_chain", lambda: mps_chain_bench(ta, be, not args.no_cpu_baseline))
      fenced(result, "helpers", lambda: helpers_bench(ta, be))
      ta.trim_pool()
      fenced(result, "gather_gemm", lambda: gather_gemm_bench(ta, be, not args.no_verify))
      if isinstance(result["gather_gemm"], dict) and "verified" in result["gather_gemm"]:
        verified["gather_gemm_equals_permute_plus_gemm"] = result["gather_gemm"].pop("verified")
      ta.trim_pool()
    if single and args.svd_n > 0:
      try:
        head, rows, chk = svd_sweep(ta, be, args.svd_n, not args.no_verify)
        result["svd"] = head
        result["svd"]["sweep"] = rows
        if chk is not None:
          verified["svd"] = chk
        try:
          wide = svd_wide_rows(ta, be, args.svd_n, not args.no_verify)
          result["svd"]["sweep"] = rows + wide
          if not args.no_verify:
            verified["svd_call_shapes"] = {"cases": {f"{r['dtype']}_{r['mode']}_{r['n']}": r["check"]["ok"] for r in wide},
                                           "paths": {f"{r['dtype']}_{r['mode']}_{r['n']}": r["path"] for r in wide},
                                           "ok": bool(all(r["check"]["ok"] for r in wide))}
        except Exception as exc:  # pylint: disable=broad-except
          result["svd"]["call_shapes_error"] = f"{type(exc).__name__}: {exc}"
        if not args.no_cpu_baseline:
          result["svd"]["cpu_baseline"] = svd_cpu_baseline(args.svd_n)
      except Exception as exc:  # pylint: disable=broad-except
        result.setdefault("svd", {})["error"] = f"{type(exc).__name__}: {exc}"
    if single and not args.no_verify and not args.no_cpu_baseline:
      fenced(verified, "mera_chi16", lambda: verify_mera(ta, be, 16))
    if single and not args.no_cpu_baseline:
      fenced(result, "cpu_baseline", lambda: cpu_baseline(args.layout))
    if verified:
      verified["all_ok"] = bool(all(v.get("ok", False) for v in verified.values() if isinstance(v, dict)))
      result["verified"] = verified
    # the JSON line must be the LAST thing on stdout: drain whatever C libraries still hold in their stdio buffers first
    sys.stdout.flush()
    try:
      import ctypes  # pylint: disable=import-outside-toplevel
      ctypes.CDLL(None).fflush(None)
    except Exception:  # pylint: disable=broad-except
      pass
    emit(result, args)
  if comm is not None:
    comm.barrier()
    comm.close()


if __name__ == "__main__":
  main()
