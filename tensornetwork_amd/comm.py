"""One process per GPU: RCCL collectives through libtnhip's C ABI (K8) + a host-side rendezvous.

``RcclComm`` is the communicator the multi-GPU paths of ``distributed.py`` and ``bench.py`` use
on GPUs.  The collectives themselves are ``tnh_allreduce`` / ``tnh_allgather`` /
``tnh_broadcast`` (``include/tnh.h``, K8): enqueued on the library's own stream, in order with the
kernels that produced the buffers -- no device-wide synchronise, no second runtime in the process.

Bootstrap needs a host channel for the 128-byte RCCL id (and for a few integers of metadata):
``HostRendezvous`` is a minimal TCP star on ``MASTER_ADDR:(MASTER_PORT + TNH_COMM_PORT_OFFSET)``
-- rank 0 listens, everyone else connects; every exchange is "all ranks send one JSON-able object,
all ranks receive the list".  The launcher contract is torchrun's environment
(``RANK`` / ``WORLD_SIZE`` / ``LOCAL_RANK`` / ``MASTER_ADDR`` / ``MASTER_PORT``); nothing here imports torch.
"""
import base64
import ctypes
import json
import os
import socket
import struct
import time

import numpy as np

from tensornetwork_amd import _lib

_PORT_OFFSET = int(os.environ.get("TNH_COMM_PORT_OFFSET", "23"))


def _send_msg(sock, obj):
  data = json.dumps(obj).encode()
  sock.sendall(struct.pack("!Q", len(data)) + data)


def _recv_exact(sock, n):
  buf = bytearray()
  while len(buf) < n:
    chunk = sock.recv(n - len(buf))
    if not chunk:
      raise ConnectionError("rendezvous peer closed the connection")
    buf.extend(chunk)
  return bytes(buf)


def _recv_msg(sock):
  (n,) = struct.unpack("!Q", _recv_exact(sock, 8))
  return json.loads(_recv_exact(sock, n).decode())


class HostRendezvous:
  """TCP star for small host-side exchanges between the ranks of one job."""

  def __init__(self, rank, world, addr=None, port=None, timeout=900.0):
    self.rank, self.world = int(rank), int(world)
    self._peers = []
    self._sock = None
    if self.world == 1:
      return
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + _PORT_OFFSET)
    if self.rank == 0:
      srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
      srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
      srv.bind((addr, port))
      srv.listen(self.world)
      srv.settimeout(timeout)
      peers = {}
      while len(peers) < self.world - 1:
        conn, _ = srv.accept()
        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        conn.settimeout(timeout)
        hello = _recv_msg(conn)
        peers[int(hello["rank"])] = conn
      srv.close()
      self._peers = [peers[r] for r in range(1, self.world)]
    else:
      deadline = time.monotonic() + timeout
      while True:
        try:
          sock = socket.create_connection((addr, port), timeout=5.0)
          break
        except OSError:
          if time.monotonic() > deadline:
            raise
          time.sleep(0.05)
      sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
      sock.settimeout(timeout)
      _send_msg(sock, {"rank": self.rank})
      self._sock = sock

  def all_gather(self, obj):
    """Every rank contributes one JSON-able object; every rank gets the list in rank order."""
    if self.world == 1:
      return [obj]
    if self.rank == 0:
      items = [obj] + [_recv_msg(p) for p in self._peers]
      for p in self._peers:
        _send_msg(p, items)
      return items
    _send_msg(self._sock, obj)
    return _recv_msg(self._sock)

  def broadcast(self, obj, root=0):
    return self.all_gather(obj if self.rank == root else None)[root]

  def barrier(self):
    self.all_gather(0)

  def close(self):
    for p in self._peers:
      try:
        p.close()
      except OSError:
        pass
    self._peers = []
    if self._sock is not None:
      try:
        self._sock.close()
      except OSError:
        pass
      self._sock = None


def single_node_rccl_env():
  """One node, rendezvous on loopback: pin RCCL's bootstrap sockets to `lo` unless the user chose an interface.
  Measured on the MI355X box (tools/rccl_two_ranks_one_gpu.py): with RCCL's own interface choice two ranks never
  get out of `ncclCommInitRank` (no error, no return within 150 s -- the sandbox has no routable interface);
  with NCCL_SOCKET_IFNAME=lo the same call answers in seconds."""
  addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
  if addr.startswith("127.") or addr in ("localhost", "::1"):
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")


def env_rank_world():
  return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
          int(os.environ.get("LOCAL_RANK", "0")))


class RcclComm:
  """Communicator over libtnhip's K8 entry points.  ``backend`` must already be bound to this
  process' GPU (``HipBackend(device=LOCAL_RANK)`` / ``$TNHIP_DEVICE``)."""

  def __init__(self, backend, rank=None, world=None, rendezvous=None):
    env_rank, env_world, _ = env_rank_world()
    self.rank = int(env_rank if rank is None else rank)
    self.world = int(env_world if world is None else world)
    self._be = backend
    self._lib = backend.lib
    single_node_rccl_env()
    self._rdv = rendezvous if rendezvous is not None else HostRendezvous(self.rank, self.world)
    try:
      self._bootstrap()
    except Exception:
      if rendezvous is None:
        self._rdv.close()
      raise

  def _bootstrap(self):
    # Lock-step bootstrap: every rank takes part in every exchange whatever happened locally, so a failure
    # on one rank raises on EVERY rank instead of leaving the others waiting -- the caller can then agree on
    # a fall-back.  ncclCommInitRank is itself collective, so anything that can fail LOCALLY (librccl does
    # not load, no device bound, a stale communicator) is checked and exchanged first (tnh_comm_available):
    # the ranks enter the collective only when all of them can.
    pre = None
    try:
      _lib.check(self._lib.tnh_comm_available(), "tnh_comm_available")
    except Exception as exc:  # pylint: disable=broad-except
      pre = f"{type(exc).__name__}: {exc}"
    unable = {r: e for r, e in enumerate(self._rdv.all_gather(pre)) if e}
    if unable:
      raise RuntimeError(f"RCCL is not usable on rank(s) {sorted(unable)}: {next(iter(unable.values()))}")
    ident, err = None, None
    if self.rank == 0:
      try:
        buf = ctypes.create_string_buffer(128)
        _lib.check(self._lib.tnh_comm_unique_id(buf), "tnh_comm_unique_id")
        ident = base64.b64encode(buf.raw).decode()
      except Exception as exc:  # pylint: disable=broad-except
        err = f"{type(exc).__name__}: {exc}"
    first = self._rdv.all_gather({"id": ident, "err": err})[0]
    if first["err"]:
      raise RuntimeError(f"RCCL id could not be created on rank 0: {first['err']}")
    err = None
    try:
      raw = ctypes.create_string_buffer(base64.b64decode(first["id"]), 128)
      _lib.check(self._lib.tnh_comm_init(raw, self.rank, self.world), "tnh_comm_init")
    except Exception as exc:  # pylint: disable=broad-except
      err = f"{type(exc).__name__}: {exc}"
    states = self._rdv.all_gather(err)
    failed = {r: e for r, e in enumerate(states) if e}
    if failed:
      if err is None:
        self._lib.tnh_comm_abort()      # never ncclCommDestroy here: it may wait for peers that never joined
      raise RuntimeError(f"tnh_comm_init failed on rank(s) {sorted(failed)}: {next(iter(failed.values()))}")

  # -- host-side metadata ------------------------------------------------------------------
  def all_gather_counts(self, n):
    return [int(x) for x in self._rdv.all_gather(int(n))]

  # -- device collectives (all on the library stream) -------------------------------------------
  def all_reduce_sum(self, backend, tensor):
    """Sum over ranks, returned in a NEW block (the argument may alias a caller-owned tensor).
    bf16 / f16 partial sums travel and add as fp32 and are rounded once at the end."""
    from tensornetwork_amd.device_tensor import DeviceTensor  # pylint: disable=import-outside-toplevel
    if not isinstance(tensor, DeviceTensor):
      raise TypeError("RcclComm reduces DeviceTensors; host arrays belong to the gloo test communicator")
    half = tensor.code in (_lib.BF16, _lib.F16)
    work = backend.cast(tensor, _lib.F32) if half else backend.copy(tensor)
    _lib.check(self._lib.tnh_allreduce(ctypes.c_void_p(work.ptr), work.size, work.code, 0), "tnh_allreduce")
    return backend.cast(work, tensor.code) if half else work

  def all_gather_rows(self, backend, tensor, rows_per_rank):
    """Concatenate the ranks' row blocks (leading axis): ONE all-gather of equal, padded blocks."""
    from tensornetwork_amd.device_tensor import DeviceTensor  # pylint: disable=import-outside-toplevel
    rows = [int(r) for r in rows_per_rank]
    pad = max(rows)
    tail = tuple(tensor.shape[1:])
    full = DeviceTensor.empty((self.world * pad,) + tail, tensor.code)
    mine = tensor
    if rows[self.rank] != pad:
      mine = DeviceTensor.empty((pad,) + tail, tensor.code)
      _lib.check(self._lib.tnh_memset(ctypes.c_void_p(mine.ptr), 0, mine.nbytes), "tnh_memset")
      backend.copy_rows_into(mine, tensor, 0)
    _lib.check(self._lib.tnh_allgather(ctypes.c_void_p(full.ptr), ctypes.c_void_p(mine.ptr), mine.nbytes),
               "tnh_allgather")
    if all(r == pad for r in rows):
      return full
    parts = [backend.getitem(full, slice(k * pad, k * pad + rows[k])) for k in range(self.world)]
    return backend.concat_rows(parts)

  def _scalar_reduce(self, value, op):
    from tensornetwork_amd.device_tensor import DeviceTensor  # pylint: disable=import-outside-toplevel
    t = DeviceTensor.from_numpy(np.asarray([float(value)], dtype=np.float64))
    _lib.check(self._lib.tnh_allreduce(ctypes.c_void_p(t.ptr), 1, _lib.F64, op), "tnh_allreduce")
    return float(t.numpy()[0])

  def max_over_ranks(self, value):
    return self._scalar_reduce(value, 1)

  def sum_over_ranks(self, value):
    return self._scalar_reduce(value, 0)

  def barrier(self):
    """Device-level barrier: a 1-element all-reduce on the library stream, then wait for it."""
    self._scalar_reduce(0.0, 0)

  def close(self):
    try:
      _lib.check(self._lib.tnh_comm_destroy(), "tnh_comm_destroy")
    finally:
      self._rdv.close()
