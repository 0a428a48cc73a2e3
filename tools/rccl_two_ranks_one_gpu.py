"""Two processes on ONE GPU try to form a 2-rank RCCL communicator through the C ABI (tools only: the GPU box
lends a single GPU).  RCCL may refuse duplicate devices; what matters here is that BOTH ranks get an answer
(communicator or RuntimeError) and nobody hangs.   python tools/rccl_two_ranks_one_gpu.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import tensornetwork_amd as ta
from tensornetwork_amd import comm, _lib
be = ta.get_hip_backend()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
try:
  c = comm.RcclComm(be, rank=rank, world=world)
  x = be.convert_to_tensor(np.full((4,), float(rank + 1), dtype=np.float32))
  y = np.asarray(c.all_reduce_sum(be, x))
  print("rank", rank, "communicator up; all_reduce_sum ->", y.tolist(), flush=True)
  c.barrier(); c.close()
except RuntimeError as exc:
  print("rank", rank, "RuntimeError:", str(exc)[:160], flush=True)
''' % ROOT
env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2", TNHIP_DEVICE="0",
           HSA_ENABLE_IPC_MODE_LEGACY="0")   # (NCCL_SOCKET_IFNAME: left to RcclComm -- tensornetwork_amd.comm.single_node_rccl_env)
procs = [subprocess.Popen([sys.executable, "-c", CHILD], env=dict(env, RANK=str(r))) for r in range(2)]
t0 = time.time()
codes = []
for p in procs:
  try:
    codes.append(p.wait(timeout=float(os.environ.get("PROBE_TIMEOUT", "150")) - (time.time() - t0)))
  except subprocess.TimeoutExpired:
    p.kill(); codes.append("timeout")
print("exit codes", codes, "seconds", round(time.time() - t0, 1))
