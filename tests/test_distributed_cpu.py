"""Bond-sliced contraction (tensornetwork_amd.distributed) on CPU: single process,
and two processes over gloo (the same code path the GPUs take over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

import tensornetwork_amd as ta
from tensornetwork_amd import contractors, distributed, network
from oracle.numpy_oracle import OracleBackend

HERE_ = os.path.dirname(os.path.abspath(__file__))
if HERE_ not in sys.path:
  sys.path.insert(0, HERE_)
from gloo_comm import GlooComm  # noqa: E402  pylint: disable=wrong-import-position

HERE = os.path.dirname(os.path.abspath(__file__))


def regular_network(be, n=8, D=3, seed=6, dtype=np.float64):
  import networkx as nx
  rng = np.random.default_rng(seed)
  g = nx.random_regular_graph(3, n, seed=seed)
  nodes = {v: network.Node(rng.standard_normal((D, D, D)).astype(dtype), backend=be) for v in sorted(g.nodes)}
  slot = {v: 0 for v in g.nodes}
  for x, y in sorted(g.edges):
    network.connect(nodes[x][slot[x]], nodes[y][slot[y]])
    slot[x] += 1
    slot[y] += 1
  return [nodes[v] for v in sorted(g.nodes)]


def test_sliced_equals_unsliced_single_process():
  be = OracleBackend()
  ref = contractors.greedy(regular_network(be)).tensor
  nodes = regular_network(be)
  cuts = distributed.choose_cut_edges(nodes, min_slices=9)
  assert len(cuts) >= 2
  rep = distributed.slicing_report(nodes, cuts)
  assert rep["n_slices"] >= 9 and rep["peak_per_slice"] <= rep["peak_unsliced"]
  out = distributed.contract_sliced(nodes, cuts)
  np.testing.assert_allclose(out, ref, rtol=1e-10)


def test_sliced_partials_are_returned_on_request():
  """bench.py's bf16-vs-f32 checks compare slice partials, not only their sum: partials_out collects them."""
  be = OracleBackend()
  nodes = regular_network(be)
  cuts = distributed.choose_cut_edges(nodes, min_slices=9)
  parts = []
  out = distributed.contract_sliced(nodes, cuts, partials_out=parts)
  assert len(parts) == int(distributed.slicing_report(nodes, cuts)["n_slices"])
  np.testing.assert_allclose(np.sum([p.reshape(-1)[0] for p in parts]), np.asarray(out).reshape(-1)[0], rtol=1e-12)


def test_sliced_with_open_edges():
  be = OracleBackend()
  rng = np.random.default_rng(0)
  a = network.Node(rng.standard_normal((3, 4, 5)), backend=be)
  b = network.Node(rng.standard_normal((5, 4, 6)), backend=be)
  e1 = a[2] ^ b[0]
  e2 = a[1] ^ b[1]
  ref = np.tensordot(a.tensor, b.tensor, [[2, 1], [0, 1]])
  out = distributed.contract_sliced([a, b], [e1], output_edge_order=[a[0], b[2]])
  np.testing.assert_allclose(out, ref, rtol=1e-12)
  with pytest.raises(ValueError):
    distributed.contract_sliced([a, b], [a[0]])
  del e2


def _worker(rank, world, port, out_path):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  sys.path.insert(0, os.path.dirname(HERE))
  sys.path.insert(0, HERE)
  import torch.distributed as dist
  dist.init_process_group(backend="gloo", rank=rank, world_size=world)
  be = OracleBackend()
  nodes = regular_network(be)
  cuts = distributed.choose_cut_edges(nodes, min_slices=9)
  comm = GlooComm()
  out = distributed.contract_sliced(nodes, cuts, comm=comm)
  np.save(out_path + f".{rank}.npy", np.asarray(out))
  dist.barrier()
  dist.destroy_process_group()


def _worker_sharded(rank, world, port, out_path):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  sys.path.insert(0, os.path.dirname(HERE))
  sys.path.insert(0, HERE)
  import torch.distributed as dist
  dist.init_process_group(backend="gloo", rank=rank, world_size=world)
  be = OracleBackend()
  rng = np.random.default_rng(3)
  a = rng.standard_normal((7, 4, 5, 6))          # 7 rows over 2 ranks: uneven blocks (4 + 3)
  b = rng.standard_normal((6, 5, 3, 2))
  comm = GlooComm()
  full, bounds = distributed.tensordot_sharded(be, a, b, [[3, 2], [0, 1]], comm=comm)
  part, pb = distributed.tensordot_sharded(be, a, b, [[3, 2], [0, 1]], comm=comm, gather=False)
  lo, hi = distributed.shard_rows(7, world)[rank]
  full2, _ = distributed.tensordot_sharded(be, a[lo:hi], b, [[3, 2], [0, 1]], comm=comm, a_is_local=True)
  np.savez(out_path + f".{rank}.npz", full=np.asarray(full), part=np.asarray(part), pb=np.asarray(pb),
           full2=np.asarray(full2))
  dist.barrier()
  dist.destroy_process_group()


def test_sharded_tensordot_two_processes_gloo(tmp_path):
  """M-sharded pairwise contraction, one all-gather of the result."""
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "res")
  mp.spawn(_worker_sharded, args=(2, port, out_path), nprocs=2, join=True)
  rng = np.random.default_rng(3)
  a = rng.standard_normal((7, 4, 5, 6))
  b = rng.standard_normal((6, 5, 3, 2))
  ref = np.tensordot(a, b, [[3, 2], [0, 1]])
  for r in range(2):
    got = np.load(out_path + f".{r}.npz")
    np.testing.assert_allclose(got["full"], ref, rtol=1e-12)
    np.testing.assert_allclose(got["full2"], ref, rtol=1e-12)
    lo, hi = got["pb"]
    assert (lo, hi) == distributed.shard_rows(7, 2)[r]
    np.testing.assert_allclose(got["part"], ref[lo:hi], rtol=1e-12)


def test_sharded_tensordot_single_process_and_errors():
  be = OracleBackend()
  rng = np.random.default_rng(4)
  a, b = rng.standard_normal((5, 6)), rng.standard_normal((6, 3))
  out, bounds = distributed.tensordot_sharded(be, a, b, 1)
  np.testing.assert_allclose(out, a @ b)
  assert bounds == (0, 5)
  assert distributed.shard_rows(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
  with pytest.raises(ValueError):
    distributed.tensordot_sharded(be, a, b, [[0], [0]])


def test_sliced_two_processes_gloo(tmp_path):
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "res")
  mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
  ref = contractors.greedy(regular_network(OracleBackend())).tensor
  for r in range(2):
    np.testing.assert_allclose(np.load(out_path + f".{r}.npy"), ref, rtol=1e-10)


def _worker_complex(rank, world, port, out_path):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  sys.path.insert(0, os.path.dirname(HERE))
  sys.path.insert(0, HERE)
  import torch.distributed as dist
  dist.init_process_group(backend="gloo", rank=rank, world_size=world)
  be = OracleBackend()
  nodes = regular_network(be, dtype=np.complex128)
  rng = np.random.default_rng(17)
  for n in nodes:   # genuinely complex entries: a dropped imaginary part changes the result
    n.tensor = n.tensor + 1j * rng.standard_normal(n.tensor.shape)
  cuts = distributed.choose_cut_edges(nodes, min_slices=9)
  out = distributed.contract_sliced(nodes, cuts, comm=GlooComm())
  np.save(out_path + f".{rank}.npy", np.asarray(out))
  dist.barrier()
  dist.destroy_process_group()


def test_sliced_complex_two_processes_gloo(tmp_path):
  """the all-reduce must not drop the imaginary part of complex partial results."""
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "res")
  mp.spawn(_worker_complex, args=(2, port, out_path), nprocs=2, join=True)
  be = OracleBackend()
  nodes = regular_network(be, dtype=np.complex128)
  rng = np.random.default_rng(17)
  for n in nodes:
    n.tensor = n.tensor + 1j * rng.standard_normal(n.tensor.shape)
  ref = contractors.greedy(nodes).tensor
  assert abs(np.imag(ref)) > 1e-6 * abs(ref)
  for r in range(2):
    np.testing.assert_allclose(np.load(out_path + f".{r}.npy"), ref, rtol=1e-10)


def test_all_reduce_never_overwrites_the_callers_tensor():
  """a trivial path (one node, no cuts) returns the node's own tensor as the partial
  result; the reduction must not write into it."""
  class Doubling(distributed.LocalComm):
    world = 2
    def all_reduce_sum(self, backend, tensor):
      return np.asarray(tensor) * 2        # what a 2-rank sum of identical partials gives, out of place
  be = OracleBackend()
  x = np.arange(6.0).reshape(2, 3)
  n = network.Node(x.copy(), backend=be)
  out = distributed.contract_sliced([n], [], comm=Doubling())
  np.testing.assert_allclose(out, 2 * x)
  np.testing.assert_allclose(n.tensor, x)


def _rdv_worker(rank, world, port, out_path):
  sys.path.insert(0, os.path.dirname(HERE))
  from tensornetwork_amd import comm
  r = comm.HostRendezvous(rank, world, addr="127.0.0.1", port=port, timeout=60)
  got = r.all_gather({"rank": rank, "n": 10 * rank})
  b = r.broadcast("id-from-root" if rank == 0 else None)
  r.barrier()
  r.close()
  with open(out_path + f".{rank}.json", "w") as f:
    import json
    json.dump({"got": got, "b": b}, f)


def test_host_rendezvous_three_processes(tmp_path):
  """The TCP star that ships the RCCL id between ranks (tensornetwork_amd.comm)."""
  import json
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "rdv")
  mp.spawn(_rdv_worker, args=(3, port, out_path), nprocs=3, join=True)
  for r in range(3):
    with open(out_path + f".{r}.json") as f:
      rec = json.load(f)
    assert rec["b"] == "id-from-root"
    assert rec["got"] == [{"rank": k, "n": 10 * k} for k in range(3)]


def _rccl_bootstrap_worker(rank, world, port, out_path):
  sys.path.insert(0, os.path.dirname(HERE))
  from tensornetwork_amd import _lib, comm

  class _NoGpuBackend:      # libtnhip loads on a CPU-only host; tnh_comm_init then fails ("not initialised")
    lib = _lib.load_library()

  rdv = comm.HostRendezvous(rank, world, addr="127.0.0.1", port=port, timeout=60)
  try:
    comm.RcclComm(_NoGpuBackend(), rank=rank, world=world, rendezvous=rdv)
    outcome = "created"
  except RuntimeError as exc:
    outcome = f"RuntimeError: {exc}"
  rdv.barrier()             # the rendezvous is still in step on every rank after the failure
  rdv.close()
  with open(out_path + f".{rank}.txt", "w") as f:
    f.write(outcome)


def test_rccl_bootstrap_failure_raises_on_every_rank(tmp_path):
  """No GPU here, so the communicator cannot come up: every rank must get the error (none may be left
  waiting in an exchange) -- bench.py relies on that to fall back to --comm torch in step."""
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "boot")
  mp.spawn(_rccl_bootstrap_worker, args=(2, port, out_path), nprocs=2, join=True)
  for r in range(2):
    with open(out_path + f".{r}.txt") as f:
      outcome = f.read()
    assert outcome.startswith("RuntimeError"), outcome
    assert "rank" in outcome


def _bench_host_comm_worker(rank, world, port, out_path):
  root = os.path.dirname(HERE)
  sys.path.insert(0, root)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TNH_COMM_PORT_OFFSET="0")
  import importlib
  import json
  from tensornetwork_amd import _lib, comm
  importlib.reload(comm)      # pick up TNH_COMM_PORT_OFFSET
  import bench

  class _NoGpuBackend:
    lib = _lib.load_library()

  c, name = bench.bring_up_comm(comm, _NoGpuBackend(), rank, world, allow_host_exchange=True)
  c.barrier()
  rec = {"type": type(c).__name__, "name": name, "rank": c.rank, "world": c.world,
         "max": c.max_over_ranks(1.5 + rank), "sum": c.sum_over_ranks(float(rank)),
         "counts": c.all_gather_counts(10 + rank),
         "f32": np.asarray(c.all_reduce_sum(None, np.array([rank + 1.0, 0.25], dtype=np.float32))).tolist(),
         "f32_dtype": str(np.asarray(c.all_reduce_sum(None, np.ones((2, 2), dtype=np.float32))).dtype),
         "c64": [[z.real, z.imag] for z in np.asarray(c.all_reduce_sum(None, np.array([1 + 1j * rank], dtype=np.complex64))).tolist()]}
  try:
    c.all_reduce_sum(None, np.zeros(bench.HostComm.MAX_ELEMS + 1, dtype=np.float32))
    rec["too_large"] = "accepted"
  except RuntimeError as exc:
    rec["too_large"] = str(exc)
  c.barrier()
  c.close()
  with open(out_path + f".{rank}.json", "w") as f:
    json.dump(rec, f)


def test_bench_host_exchange_when_rccl_cannot_come_up(tmp_path):
  """bench.bring_up_comm with --allow-host-exchange: no GPU here, so the K8 communicator raises on every rank -- and every rank then holds a
  HostComm (TCP rendezvous): barrier, max / sum of a scalar, the small all-reduce of the sliced network (f32, complex,
  the same bits on every rank), a size limit.  `config.comm` names the host exchange and the reason."""
  import json
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "hc")
  world = 3
  mp.spawn(_bench_host_comm_worker, args=(world, port, out_path), nprocs=world, join=True)
  for r in range(world):
    with open(out_path + f".{r}.json") as f:
      rec = json.load(f)
    assert rec["type"] == "HostComm" and (rec["rank"], rec["world"]) == (r, world)
    assert rec["name"].startswith("HOST EXCHANGE") and "RCCL is not usable" in rec["name"]
    assert rec["max"] == 1.5 + (world - 1) and rec["sum"] == float(sum(range(world)))
    assert rec["counts"] == [10 + k for k in range(world)]
    assert rec["f32"] == [float(sum(k + 1 for k in range(world))), 0.25 * world] and rec["f32_dtype"] == "float32"
    assert rec["c64"] == [[float(world), float(sum(range(world)))]]
    assert "small results" in rec["too_large"]


def _bench_no_rccl_worker(rank, world, port, out_path):
  root = os.path.dirname(HERE)
  sys.path.insert(0, root)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TNH_COMM_PORT_OFFSET="0")
  import importlib
  from tensornetwork_amd import _lib, comm
  importlib.reload(comm)
  import bench

  class _NoGpuBackend:
    lib = _lib.load_library()

  try:
    bench.bring_up_comm(comm, _NoGpuBackend(), rank, world)          # the default: no host exchange
    outcome = "came up"
  except SystemExit as exc:
    outcome = f"SystemExit {exc.code}"
  with open(out_path + f".{rank}.txt", "w") as f:
    f.write(outcome)


def test_bench_exits_nonzero_when_rccl_cannot_come_up_by_default(tmp_path):
  """without --allow-host-exchange a multi-rank bench whose RCCL communicator does not come up
  ENDS (exit code bench.EXIT_NO_RCCL, on every rank, in step) instead of printing an N-rank line without RCCL."""
  import torch.multiprocessing as mp
  import bench
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "norccl")
  mp.spawn(_bench_no_rccl_worker, args=(2, port, out_path), nprocs=2, join=True)
  for r in range(2):
    with open(out_path + f".{r}.txt") as f:
      assert f.read() == f"SystemExit {bench.EXIT_NO_RCCL}"
  assert bench.EXIT_NO_RCCL != 0
  assert bench.parse_args(["--gpus", "2"]).allow_host_exchange is False
  assert bench.parse_args(["--gpus", "2", "--allow-host-exchange"]).allow_host_exchange is True


def test_bench_line_carries_rccl_ranks_and_the_strong_scaling_claim():
  """the compact line says how many ranks the LIVE RCCL communicator has (0 for the host exchange
  or a single process) and carries the strong-scaling numbers of the sliced network -- N-GPU seconds, the same cuts
  on one GPU, speedup, the partition's ideal -- so the >= 6x claim needs no detail file."""
  import json
  import bench
  sn = {"workload": "64-node random 3-regular network (seed 6), bond D=16, bf16, 2 cut bonds", "n_slices": 256, "n_gpus": 8,
        "seconds": 0.2, "seconds_1gpu_same_cuts": 1.4, "speedup_over_1gpu_same_cuts": 7.0, "ideal_speedup_of_this_partition": 7.99,
        "tflops": 6500.0, "scaling": "strong"}
  base = {"metric": "m", "value": 1.0, "unit": "TFLOP/s", "n_gpus": 8, "scaling": "weak", "config": {"comm": bench.RCCL_COMM_NAME},
          "sliced_network": sn}
  line = json.loads(bench.compact_line(dict(base, rccl_ranks=8), "d.json"))
  assert line["rccl_ranks"] == 8 and line["scaling"] == "weak"
  assert line["strong_scaling"] == {"workload": sn["workload"][:72], "n_gpus": 8, "seconds": 0.2, "seconds_1gpu_same_cuts": 1.4,
                                    "speedup": 7.0, "ideal": 7.99, "collective": "rccl"}
  # the labelled host exchange: no RCCL ranks, and the strong-scaling entry says whose exchange it was
  line = json.loads(bench.compact_line(dict(base, rccl_ranks=0), "d.json"))
  assert line["rccl_ranks"] == 0 and line["strong_scaling"]["collective"] == "HOST"
  one = dict(base, n_gpus=1, rccl_ranks=0, sliced_network=dict(sn, n_gpus=1, seconds=1.4, speedup_over_1gpu_same_cuts=1.0))
  line = json.loads(bench.compact_line(one, "d.json"))
  assert line["strong_scaling"]["collective"] == "none" and line["strong_scaling"]["speedup"] == 1.0

  class _Lib:
    def tnh_comm_info(self, r, w):
      r._obj.value, w._obj.value = 3, 8       # pylint: disable=protected-access
      return 0

  class _Be:
    lib = _Lib()
  assert bench.rccl_ranks(_Be(), object()) == 8
  assert bench.rccl_ranks(_Be(), None) == 0
  assert bench.rccl_ranks(_Be(), bench.HostComm.__new__(bench.HostComm)) == 0


def test_single_node_rccl_env_pins_loopback(monkeypatch):
  """One node (MASTER_ADDR on loopback): RCCL's bootstrap sockets are pinned to `lo` unless the user chose an
  interface; a routable MASTER_ADDR is left alone."""
  from tensornetwork_amd import comm
  monkeypatch.delenv("NCCL_SOCKET_IFNAME", raising=False)
  monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
  comm.single_node_rccl_env()
  assert os.environ["NCCL_SOCKET_IFNAME"] == "lo"
  monkeypatch.setenv("NCCL_SOCKET_IFNAME", "eth7")
  comm.single_node_rccl_env()
  assert os.environ["NCCL_SOCKET_IFNAME"] == "eth7"
  monkeypatch.delenv("NCCL_SOCKET_IFNAME")
  monkeypatch.setenv("MASTER_ADDR", "10.1.2.3")
  comm.single_node_rccl_env()
  assert "NCCL_SOCKET_IFNAME" not in os.environ


def _bench_comm_worker(rank, world, port, out_path):
  root = os.path.dirname(HERE)
  sys.path.insert(0, root)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TNH_COMM_PORT_OFFSET="0")
  import importlib
  from tensornetwork_amd import _lib, comm
  importlib.reload(comm)      # pick up TNH_COMM_PORT_OFFSET
  import bench

  class _NoGpuBackend:
    lib = _lib.load_library()

  try:
    bench.make_rccl_comm(comm, _NoGpuBackend(), rank, world)
    outcome = "created"
  except RuntimeError as exc:
    outcome = f"RuntimeError: {exc}"
  with open(out_path + f".{rank}.txt", "w") as f:
    f.write(outcome)


def test_bench_has_one_communicator_and_its_failure_raises_on_every_rank(tmp_path):
  """bench.make_rccl_comm: when the K8 communicator cannot come up (no GPU here) EVERY rank gets the exception in
  step (none hangs in an exchange); the package has ONE communicator and no torch.distributed route (VERDICT r3
  weak 9).  What bench.py does with that exception: test_bench_host_exchange_when_rccl_cannot_come_up."""
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "fb")
  mp.spawn(_bench_comm_worker, args=(2, port, out_path), nprocs=2, join=True)
  for r in range(2):
    with open(out_path + f".{r}.txt") as f:
      assert f.read().startswith("RuntimeError: RCCL is not usable on rank(s) [0, 1]")
  root = os.path.dirname(HERE)
  for rel in ("bench.py", "tensornetwork_amd/distributed.py", "tensornetwork_amd/comm.py"):
    src = open(os.path.join(root, rel)).read()
    assert "import torch" not in src and "torch.distributed" not in src.replace("torch.distributed.run", ""), rel


def test_bench_watchdog_turns_a_hang_into_exit_code_3(tmp_path):
  """a step that hangs inside a collective must end the rank with a message, not the record with a
  driver-side timeout.  The watchdog fires while the main thread is blocked (here: in a sleep), and stays quiet when
  the block finishes in time."""
  import subprocess
  root = os.path.dirname(HERE)
  code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
          "with bench.Watchdog(30, 'fast step', 0):\n  pass\n"
          "with bench.Watchdog(0.5, 'RCCL bring-up', 5):\n  time.sleep(60)\n"
          "print('not reached')\n" % root)
  t0 = __import__("time").monotonic()
  res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
  assert res.returncode == 3 and __import__("time").monotonic() - t0 < 30
  assert "rank 5: RCCL bring-up did not finish within" in res.stderr and "not reached" not in res.stdout


def test_self_launch_stops_every_rank_when_one_fails_or_the_job_times_out(tmp_path, monkeypatch):
  """bench.self_launch watches its ranks: one failing rank takes the others down (they would wait in a collective
  for ever), and the job as a whole has a wall-clock limit.  Ranks here: a stand-in script instead of bench.py."""
  import argparse
  import bench
  script = tmp_path / "rank.py"
  script.write_text("import os, sys, time\n"
                    "mode = os.environ['FAKE_MODE']\n"
                    "if mode == 'fail' and os.environ['RANK'] == '1':\n  sys.exit(7)\n"
                    "time.sleep(120)\n")
  monkeypatch.setattr(bench, "__file__", str(script))
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
  for mode, timeout in (("fail", 100.0), ("hang", 1.0)):
    monkeypatch.setenv("FAKE_MODE", mode)
    args = argparse.Namespace(gpus=3, dry_run=True, job_timeout=timeout)
    t0 = __import__("time").monotonic()
    assert bench.self_launch(args) == 1
    assert __import__("time").monotonic() - t0 < 30, mode


def test_sliced_result_dtype_follows_the_per_slice_result():
  """the final cast used to follow nodes[0]'s dtype.  The rule now: round back only what was widened
  -- a half per-slice result -- so half_output="float32" backends and mixed networks keep their wide result."""

  class _T:
    def __init__(self, dtype):
      self.dtype = dtype

  class _Be:
    def cast(self, t, dtype):
      return _T(np.dtype(dtype) if not isinstance(dtype, str) else dtype)

  class _Comm:
    def all_reduce_sum(self, backend, tensor):
      return tensor

  be = _Be()
  wide, narrow = distributed._widen(be, _T("bfloat16"))
  assert str(wide.dtype) == "float32" and narrow == "bfloat16"
  assert distributed._finish(be, _Comm(), wide, narrow).dtype == "bfloat16"
  # per-slice result already fp32 (half_output="float32", or a mixed network): nothing is narrowed afterwards
  wide, narrow = distributed._widen(be, _T(np.dtype(np.float32)))
  assert narrow is None
  assert distributed._finish(be, _Comm(), wide, narrow).dtype == np.float32


def _asym_bootstrap_worker(rank, world, port, out_path):
  sys.path.insert(0, os.path.dirname(HERE))
  from tensornetwork_amd import comm

  class _Lib:
    """rank 1 cannot load RCCL; rank 0 could -- and would block inside the collective init for good"""
    def __init__(self, rank):
      self.rank, self.entered_init = rank, False
    def tnh_comm_available(self):
      return -3 if self.rank == 1 else 0
    def tnh_last_error(self):
      return b"cannot load librccl (mock)"
    def tnh_comm_unique_id(self, buf):
      return 0
    def tnh_comm_init(self, raw, rank, world):
      self.entered_init = True
      return 0
    def tnh_comm_abort(self):
      return 0

  class _Be:
    pass
  be = _Be()
  be.lib = _Lib(rank)
  rdv = comm.HostRendezvous(rank, world, addr="127.0.0.1", port=port, timeout=60)
  try:
    comm.RcclComm(be, rank=rank, world=world, rendezvous=rdv)
    outcome = "created"
  except RuntimeError as exc:
    outcome = f"RuntimeError: {exc}"
  rdv.barrier()
  rdv.close()
  with open(out_path + f".{rank}.txt", "w") as f:
    f.write(f"{outcome}|entered_init={be.lib.entered_init}")


def test_rccl_bootstrap_asymmetric_failure_never_enters_the_collective(tmp_path):
  """one rank failing BEFORE ncclCommInitRank used to leave the healthy ranks inside it.  The
  pre-flight exchange (tnh_comm_available) makes every rank raise without any of them entering the init."""
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "asym")
  mp.spawn(_asym_bootstrap_worker, args=(2, port, out_path), nprocs=2, join=True)
  for r in range(2):
    with open(out_path + f".{r}.txt") as f:
      outcome = f.read()
    assert outcome.startswith("RuntimeError") and "rank(s) [1]" in outcome, outcome
    assert outcome.endswith("entered_init=False"), outcome


def test_bench_gpus_n_launches_n_ranks_or_fails_loudly():
  """`python bench.py --gpus 2` without a launcher used to benchmark ONE GPU and print
  n_gpus: 1.  Now it starts the ranks itself (dry run: rendezvous only) and refuses when devices are missing."""
  import json
  import subprocess
  root = os.path.dirname(HERE)
  env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
  dry = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
  assert dry.returncode == 0, dry.stderr
  rec = json.loads(dry.stdout.strip().splitlines()[-1])
  assert rec["n_gpus"] == 2 and rec["ranks_seen"] == [0, 1] and rec["distinct_processes"] == 2
  # no GPU on this host: the real run must exit non-zero with a message, not fall back to one device
  env["HIP_VISIBLE_DEVICES"] = ""
  real = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                        capture_output=True, text=True, timeout=300)
  assert real.returncode != 0 and "n_gpus" not in real.stdout
  assert "--gpus 2" in real.stderr
  # a launcher that disagrees with the flag is an error as well
  env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
  odd = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env2,
                       capture_output=True, text=True, timeout=300)
  assert odd.returncode != 0 and "n_gpus" not in odd.stdout


def _worker_rr64(rank, world, port, out_path):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  sys.path.insert(0, os.path.dirname(HERE))
  sys.path.insert(0, HERE)
  import torch
  import torch.distributed as dist
  torch.set_num_threads(1)
  dist.init_process_group(backend="gloo", rank=rank, world_size=world)
  from tensornetwork_amd import workloads
  be = OracleBackend()
  nodes = workloads.random_regular_network(be, n=64, D=4, seed=6, dtype=np.float64)
  cuts = distributed.choose_cut_edges(nodes, min_slices=64)
  rep = distributed.slicing_report(nodes, cuts)
  mine = []
  out = distributed.contract_sliced(nodes, cuts, comm=GlooComm(), partials_out=mine)
  np.savez(out_path + f".{rank}.npz", out=np.asarray(out), n_mine=len(mine), n_slices=int(rep["n_slices"]))
  dist.barrier()
  dist.destroy_process_group()


def test_eight_rank_rehearsal_of_the_north_star_network(tmp_path):
  """the first 8-rank run must not be the first time the 8-rank logic executes.  The bench's own
  workload -- the 64-node random 3-regular network of SURVEY 8d (graph seed 6), here at D = 4 on the oracle backend
  over gloo -- on 8 ranks: the launcher starts 8 ranks that find each other (`--gpus 8 --dry-run`), every rank gets
  its share of the slices (round-robin, within one of each other), and the 8-rank all-reduced value equals the
  1-rank value."""
  import json
  import subprocess
  import torch.multiprocessing as mp
  root = os.path.dirname(HERE)
  env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
  dry = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=600)
  assert dry.returncode == 0, dry.stderr
  rec = json.loads(dry.stdout.strip().splitlines()[-1])
  assert rec["n_gpus"] == 8 and rec["ranks_seen"] == list(range(8)) and rec["distinct_processes"] == 8
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  out_path = str(tmp_path / "rr")
  mp.spawn(_worker_rr64, args=(8, port, out_path), nprocs=8, join=True)
  from tensornetwork_amd import workloads
  be = OracleBackend()
  nodes = workloads.random_regular_network(be, n=64, D=4, seed=6, dtype=np.float64)
  cuts = distributed.choose_cut_edges(nodes, min_slices=64)
  ref = np.asarray(distributed.contract_sliced(nodes, cuts))
  got = [np.load(out_path + f".{r}.npz") for r in range(8)]
  n_slices = int(got[0]["n_slices"])
  assert n_slices >= 64 and sum(int(g["n_mine"]) for g in got) == n_slices
  assert max(int(g["n_mine"]) for g in got) - min(int(g["n_mine"]) for g in got) <= 1
  for g in got:
    np.testing.assert_allclose(g["out"], ref, rtol=1e-9)


def test_slice_invariant_steps_are_contracted_once():
  """contract_sliced runs the steps of the path that touch no cut bond ONCE, before the slice loop: same result and
  the same slice partials as without hoisting, fewer pairwise contractions, and slicing_report counts the same steps."""
  from tensornetwork_amd import network as net_mod  # pylint: disable=import-outside-toplevel
  nodes = regular_network(OracleBackend(), n=14, D=3, seed=4)
  cuts = distributed.choose_cut_edges(nodes, min_slices=9)
  calls = {"n": 0}
  real = net_mod.contract_between

  def counting(*args, **kwargs):
    calls["n"] += 1
    return real(*args, **kwargs)

  results = {}
  net_mod.contract_between = counting
  try:
    for on in (True, False):
      calls["n"], parts, stats = 0, [], {}
      out = distributed.contract_sliced(nodes, cuts, hoist_invariant=on, partials_out=parts, stats=stats, reuse=False)
      results[on] = (np.asarray(out), [np.asarray(p) for p in parts], calls["n"], dict(stats))
  finally:
    net_mod.contract_between = real
  rep = distributed.slicing_report(nodes, cuts)
  n_slices = int(rep["n_slices"])
  (out1, parts1, calls1, stats1), (out0, parts0, calls0, stats0) = results[True], results[False]
  np.testing.assert_allclose(out1, out0, rtol=1e-12, atol=1e-12)
  assert len(parts1) == len(parts0) == n_slices
  for a, b in zip(parts1, parts0):
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
  assert stats0["hoisted_steps"] == 0 and stats1["hoisted_steps"] == rep["invariant_steps"] > 0
  assert stats1["steps_per_slice"] == rep["steps_per_slice"] - rep["invariant_steps"]
  assert calls0 == n_slices * rep["steps_per_slice"]
  assert calls1 == rep["invariant_steps"] + n_slices * stats1["steps_per_slice"]
  assert 0 < rep["flops_invariant_per_slice"] < rep["flops_per_slice"]
  # the unsliced contraction of the same network
  ref = contractors.greedy(list(network.copy(nodes)[0].values())).tensor
  np.testing.assert_allclose(out1, np.asarray(ref), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("seed,n,min_slices", [(4, 14, 9), (6, 12, 27), (9, 16, 4)])
def test_steps_run_once_per_value_of_the_cuts_they_depend_on(seed, n, min_slices):
  """Staged reuse (the default when it saves a fifth of the work): a step that depends on the cut bonds S runs once per
  distinct value tuple of S.  Same result, the same multiset of slice partials as the slice-by-slice run, and exactly
  the predicted number of stage runs and multiply-adds."""
  nodes = regular_network(OracleBackend(), n=n, D=3, seed=seed)
  cuts = distributed.choose_cut_edges(nodes, min_slices=min_slices)
  parts_s, parts_0, stats_s, stats_0 = [], [], {}, {}
  out_s = distributed.contract_sliced(nodes, cuts, reuse=True, partials_out=parts_s, stats=stats_s)
  out_0 = distributed.contract_sliced(nodes, cuts, reuse=False, hoist_invariant=False, partials_out=parts_0, stats=stats_0)
  np.testing.assert_allclose(np.asarray(out_s), np.asarray(out_0), rtol=1e-10, atol=1e-12)
  assert stats_s["mode"] == "staged" and stats_0["mode"] == "slice by slice"
  a = np.sort(np.array([float(np.asarray(p).reshape(-1)[0]) for p in parts_s]))
  b = np.sort(np.array([float(np.asarray(p).reshape(-1)[0]) for p in parts_0]))
  np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)
  # the plan predicts what ran
  inputs, output, sizes = distributed._index_problem(nodes)      # pylint: disable=protected-access
  sliced = dict(sizes)
  for e in cuts:
    sliced[e] = 1
  from tensornetwork_amd import pathfinder  # pylint: disable=import-outside-toplevel
  plan = distributed._StagePlan(nodes, cuts, pathfinder.greedy(inputs, output, sliced))      # pylint: disable=protected-access
  dims = [e.dimension for e in cuts]
  for c, runs in stats_s["stage_runs"].items():
    ks = [int(x) for x in c.split(",")] if c != "-" else []
    assert runs == int(np.prod([dims[k] for k in ks])) if ks else runs == 1
  n_slices = int(np.prod(dims))
  assert stats_s["executed_macs"] == sum(m * int(np.prod([dims[k] for k in c])) for c, m in plan.class_macs.items())
  assert stats_s["executed_macs"] <= plan.macs_alone() * n_slices
  assert abs(plan.macs_alone() - distributed.slicing_report(nodes, cuts)["flops_per_slice"]) < 1e-6 * plan.macs_alone()


def test_staged_reuse_with_open_edges_and_ranks():
  """Open legs come out in the requested order in the staged mode too, and contiguous blocks of the reordered slices
  over the ranks add up to the whole."""
  be = OracleBackend()
  rng = np.random.default_rng(3)
  a = network.Node(rng.standard_normal((3, 4, 5)), backend=be)
  b = network.Node(rng.standard_normal((4, 5, 6, 2)), backend=be)
  c = network.Node(rng.standard_normal((2, 6, 7)), backend=be)
  e1 = network.connect(a[1], b[0])
  network.connect(a[2], b[1])
  network.connect(b[2], c[1])
  e2 = network.connect(b[3], c[0])
  ref = np.einsum("xab,abcd,dcy->yx", np.asarray(a.tensor), np.asarray(b.tensor), np.asarray(c.tensor))
  out = distributed.contract_sliced([a, b, c], [e1, e2], output_edge_order=[c[2], a[0]], reuse=True)
  np.testing.assert_allclose(np.asarray(out), ref, rtol=1e-10)

  class Rank(distributed.LocalComm):
    def __init__(self, rank, world):
      self.rank, self.world = rank, world

  total = sum(np.asarray(distributed.contract_sliced([a, b, c], [e1, e2], comm=Rank(r, 3), output_edge_order=[c[2], a[0]],
                                                     reuse=True)) for r in range(3))
  np.testing.assert_allclose(total, ref, rtol=1e-10)
  # more ranks than slices: the idle ranks contribute zeros of the right shape
  outs = [np.asarray(distributed.contract_sliced([a, b, c], [e2], comm=Rank(r, 3), output_edge_order=[c[2], a[0]], reuse=True))
          for r in range(3)]
  assert outs[2].shape == ref.shape and not outs[2].any()
  np.testing.assert_allclose(sum(outs), ref, rtol=1e-10)


def test_cut_selection_by_beam_search_is_never_worse_in_estimated_time():
  """choose_cut_edges ranks cut SETS by the estimated time of what the staged contraction executes (`step_seconds`
  per pairwise step, every step once per distinct value of the cuts it depends on): the beam search never does worse
  there than the sequential rule, the default (beam 24, the better of the two) never worse than either, and the
  contraction agrees."""
  import itertools  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import pathfinder  # pylint: disable=import-outside-toplevel
  for seed in (2, 4, 7):
    nodes = regular_network(OracleBackend(), n=14, D=3, seed=seed)

    def estimated(cuts):
      inputs, output, sizes = distributed._index_problem(nodes)      # pylint: disable=protected-access
      sliced = dict(sizes)
      for e in cuts:
        sliced[e] = 1
      plan = distributed._StagePlan(nodes, list(cuts), pathfinder.greedy(inputs, output, sliced))      # pylint: disable=protected-access
      return plan.seconds_with_reuse(list(itertools.product(*[range(e.dimension) for e in cuts])))

    seq = distributed.choose_cut_edges(nodes, min_slices=9, beam=0)     # the sequential rule alone (rounds 1-4)
    beam = distributed.choose_cut_edges(nodes, min_slices=9, beam=8)
    default = distributed.choose_cut_edges(nodes, min_slices=9)         # round 5: beam 24, the better of the two
    assert int(np.prod([e.dimension for e in beam])) >= 9
    assert estimated(beam) <= estimated(seq) * (1 + 1e-12)
    assert estimated(default) <= estimated(beam) * (1 + 1e-12)
    np.testing.assert_allclose(np.asarray(distributed.contract_sliced(nodes, beam)), np.asarray(distributed.contract_sliced(nodes, seq)),
                               rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("seed,n", [(1, 10), (4, 12), (6, 14)])
def test_default_mode_decision_is_the_same_on_every_rank(seed, n):
  """with `reuse=None` the staged / slice-by-slice decision was taken from each rank's OWN block;
  with a short or empty last block (9 slices on 4 / 5 / 7 / 8 ranks, 27 on 8) ranks chose differently, took different
  shares of the slices and the all-reduced sum was wrong without any error.  The decision now comes from all ranks'
  blocks: whatever the world size, every rank reports the same mode and the shares add up to the one-rank result."""
  be = OracleBackend()
  nodes = regular_network(be, n=n, D=3, seed=seed)
  ref = float(np.asarray(contractors.greedy(regular_network(be, n=n, D=3, seed=seed)).tensor))

  class Rank(distributed.LocalComm):
    def __init__(self, rank, world):
      self.rank, self.world = rank, world

  for min_slices in (9, 27):
    cuts = distributed.choose_cut_edges(nodes, min_slices=min_slices)
    n_slices = int(np.prod([e.dimension for e in cuts]))
    for world in (3, 4, 5, 7, 8):
      modes, total, counted = set(), 0.0, 0
      for r in range(world):
        st = {}
        total += float(np.asarray(distributed.contract_sliced(nodes, cuts, comm=Rank(r, world), stats=st)))
        modes.add(st["mode"])
        counted += st["slices"]
      assert len(modes) == 1, (min_slices, world, modes)
      assert counted == n_slices, (min_slices, world, counted)
      assert abs(total - ref) <= 1e-9 * max(abs(ref), 1e-6), (min_slices, world, total, ref)


def test_partition_minimises_the_slowest_rank_and_is_rank_independent():
  """`_StagePlan.partition`: pure host arithmetic (the same blocks whoever asks), every slice exactly once, and the
  slowest rank never worse than under the round-4 rule (ceil-sized contiguous blocks of the weight order)."""
  import itertools  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import pathfinder  # pylint: disable=import-outside-toplevel
  be = OracleBackend()
  nodes = regular_network(be, n=14, D=3, seed=2)
  cuts = distributed.choose_cut_edges(nodes, min_slices=27)
  inputs, output, sizes = distributed._index_problem(nodes)      # pylint: disable=protected-access
  for e in cuts:
    sizes[e] = 1
  plan = distributed._StagePlan(nodes, cuts, pathfinder.greedy(inputs, output, sizes))      # pylint: disable=protected-access
  every = list(itertools.product(*[range(e.dimension) for e in cuts]))
  for world in (1, 2, 3, 5, 8, 40):
    blocks = plan.partition(every, world)
    assert len(blocks) == world and sorted(x for b in blocks for x in b) == sorted(every)
    assert blocks == plan.partition(list(reversed(every)), world)
    ordered = plan.ordered(every)
    per = -(-len(ordered) // world)
    old = [ordered[r * per:(r + 1) * per] for r in range(world)]
    assert max(plan.seconds_with_reuse(b) for b in blocks) <= max(plan.seconds_with_reuse(b) for b in old) * (1 + 1e-12)


def test_partition_takes_a_grid_where_two_single_cut_classes_cost_something():
  """The chi = 64 MERA placement (shapes only): the classes that depend on ONE cut are 1.5 % of the work each way, a rank
  of 8 that owns 8 values of the slow cut and all 64 of the fast one repeats the fast cut's class 64 times (ideal 7.60x,
  64 results of 2 GB kept); the 4 x 2 grid repeats 16 + 32 (ideal 7.71x, 32 results kept).  `partition` must find it, keep
  every slice exactly once, and stay with contiguous blocks where a grid is not strictly better (4 ranks: a tie)."""
  import itertools  # pylint: disable=import-outside-toplevel
  from tensornetwork_amd import workloads  # pylint: disable=import-outside-toplevel
  plan = workloads._mera_slice_plan(64, "left")["stage"]      # pylint: disable=protected-access
  every = plan.ordered(list(itertools.product(range(64), range(64))))
  ideal = lambda blocks: plan.macs_with_reuse(every) / max(plan.macs_with_reuse(b) for b in blocks)
  b8 = plan.partition(every, 8)
  assert sorted(x for b in b8 for x in b) == sorted(every) and [len(b) for b in b8] == [512] * 8
  assert sorted({len({i for i, _ in b}) for b in b8}) == [16] and sorted({len({j for _, j in b}) for b in b8}) == [32]
  assert 7.70 < ideal(b8) < 7.72
  b4 = plan.partition(every, 4)
  assert all(len({i for i, _ in b}) == 16 and len({j for _, j in b}) == 64 for b in b4)      # contiguous blocks (tie)
  # a subset of the slices (not the full product): contiguous blocks only
  sub = [s for s in every if s[0] < 5]
  bs = plan.partition(sub, 8)
  assert sorted(x for b in bs for x in b) == sorted(sub)
  assert plan._grids(8) == [(1, 8), (2, 4), (4, 2), (8, 1)]      # pylint: disable=protected-access


def test_staged_contraction_releases_what_it_kept_when_it_returns():
  """`stage` in `_contract_slices_staged` is a closure that refers to itself: the results it kept would wait for the
  cyclic collector (up to 60 % of the HBM; the next call would size its cache on the rest -- seen in the 8-rank
  rehearsal of the chi = 64 MERA placement, profiles/r05_mera_rank_share_rehearsal.jsonl).  They are dropped
  explicitly: with the collector switched off, the tensors a call kept are gone when it has returned."""
  import gc  # pylint: disable=import-outside-toplevel
  import weakref  # pylint: disable=import-outside-toplevel
  be = OracleBackend()
  nodes = regular_network(be, n=10, D=3, seed=4)
  cuts = distributed.choose_cut_edges(nodes, min_slices=9)
  seen = []
  orig = contractors.contract_labelled

  class Arr(np.ndarray):      # ndarray subclass: weak-referenceable
    pass

  def spy(be_, operands, steps, label_time=None):
    out = orig(be_, operands, steps, label_time)
    out = {k: (np.asarray(t).view(Arr), labs) for k, (t, labs) in out.items()}
    seen.extend(weakref.ref(t) for t, _ in out.values())
    return out
  gc.collect()
  gc.disable()
  try:
    contractors.contract_labelled = spy
    want = distributed.contract_sliced(nodes, cuts, reuse=True)
    alive = sum(1 for r in seen if r() is not None)
  finally:
    contractors.contract_labelled = orig
    gc.enable()
  assert seen and alive <= 1, (alive, len(seen))            # at most the tensor the result is a view of
  np.testing.assert_allclose(np.asarray(want), np.asarray(distributed.contract_sliced(nodes, cuts, reuse=False)), rtol=1e-10)


def test_cut_selection_prefers_the_faster_set_not_the_fewest_multiply_adds():
  """Round-5 measurement (profiles/r05_rr_scaling_rehearsal.jsonl): on the D = 12 64-node network the cut pair with
  the fewest executed multiply-adds (1.73e13) ran 3.4x slower on the GPU than the sequential rule's pair (2.18e13):
  its work sits in thin per-slice products.  The selection therefore ranks by `step_seconds`; on that network it must
  keep the sequential pair for 1 and for 8 ranks (checked on shapes only: nothing is contracted)."""
  from tensornetwork_amd import workloads  # pylint: disable=import-outside-toplevel
  be = OracleBackend()
  nodes = workloads.random_regular_network(be, n=64, D=12, seed=6, tensors=[np.zeros((12, 12, 12), dtype=np.float32)] * 64)
  seq = distributed.choose_cut_edges(nodes, min_slices=64, beam=0)
  for world in (1, 8):
    got = distributed.choose_cut_edges(nodes, min_slices=64, world=world)
    assert [id(e) for e in got] == [id(e) for e in seq]
  rep8 = distributed.slicing_report(nodes, seq, world=8)
  assert rep8["slices_per_rank"] == [18] * 8 and rep8["staged_by_default"]
  # D = 16: the costly cut's 16 values divide evenly among 8 ranks
  nodes16 = workloads.random_regular_network(be, n=64, D=16, seed=6, tensors=[np.zeros((16, 16, 16), dtype=np.float32)] * 64)
  cuts16 = distributed.choose_cut_edges(nodes16, min_slices=64, world=8)
  r1, r8 = distributed.slicing_report(nodes16, cuts16, world=1), distributed.slicing_report(nodes16, cuts16, world=8)
  assert r1["flops_with_reuse_slowest_rank"] / r8["flops_with_reuse_slowest_rank"] >= 7.9
  assert r1["model_seconds_slowest_rank"] / r8["model_seconds_slowest_rank"] >= 7.5
