"""TEST INFRASTRUCTURE: a communicator over an initialised ``torch.distributed`` (gloo) process group on HOST arrays.

The product package has ONE communicator -- ``tensornetwork_amd.comm.RcclComm`` (RCCL through libtnhip's C ABI).  The
partitioning logic of ``tensornetwork_amd.distributed`` only needs an object with ``rank`` / ``world`` /
``all_reduce_sum`` / ``all_gather_rows`` / ``all_gather_counts``; the CPU suite (world_size 2 and 3, oracle backend)
supplies this one.  Nothing under ``tensornetwork_amd/`` imports torch.
"""
import numpy as np


class GlooComm:
  """NumPy tensors (oracle backend) summed / gathered through gloo."""

  def __init__(self):
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel
    if not dist.is_initialized():
      raise RuntimeError("torch.distributed process group is not initialised")
    self._dist = dist
    self.rank = dist.get_rank()
    self.world = dist.get_world_size()

  def all_gather_counts(self, n):
    """Every rank's integer (host-side metadata exchange)."""
    outs = [None] * self.world
    self._dist.all_gather_object(outs, int(n))
    return [int(x) for x in outs]

  def all_reduce_sum(self, backend, tensor):  # pylint: disable=unused-argument
    import torch  # pylint: disable=import-outside-toplevel
    host = np.ascontiguousarray(np.asarray(tensor))
    flat = host.reshape(-1).copy()
    if flat.dtype.kind == "c":   # gloo has no complex sum: reduce the interleaved real image
      real = flat.view(np.float32 if flat.dtype == np.complex64 else np.float64)
      t = torch.from_numpy(real)
      self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
      return t.numpy().view(flat.dtype).reshape(host.shape)
    t = torch.from_numpy(flat)
    self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
    return t.numpy().reshape(host.shape)

  def all_gather_rows(self, backend, tensor, rows_per_rank):  # pylint: disable=unused-argument
    """Concatenate the ranks' row blocks (leading axis) of a sharded result: ONE all-gather of equal, padded blocks;
    ``rows_per_rank`` lists every rank's true row count."""
    import torch  # pylint: disable=import-outside-toplevel
    rows = [int(r) for r in rows_per_rank]
    pad = max(rows)
    host = np.ascontiguousarray(np.asarray(tensor))
    tail = tuple(host.shape[1:])
    buf = np.zeros((pad,) + tail, dtype=host.dtype)
    buf[:rows[self.rank]] = host
    src = torch.from_numpy(buf.view(np.uint8).reshape(-1).copy())
    outs = [torch.empty_like(src) for _ in range(self.world)]
    self._dist.all_gather(outs, src)
    blocks = [o.numpy().view(host.dtype).reshape((pad,) + tail)[:rows[k]] for k, o in enumerate(outs)]
    return np.concatenate(blocks, axis=0)
