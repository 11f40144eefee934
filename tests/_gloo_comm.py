"""TEST-ONLY transport for `spectralcluster_amd.multigpu.Comm`: torch.distributed with
the gloo backend, so that the partitioning logic can run with world size 2 on CPU (and with
two ranks sharing the one GPU of a test box, which RCCL refuses).  The product path uses
`multigpu.RcclComm` (RCCL behind the C ABI) and never imports torch."""

import numpy as np

from spectralcluster_amd import multigpu


class GlooComm(multigpu.Comm):

  def __init__(self):
    import torch.distributed as dist
    self._dist = dist
    self.rank, self.size = dist.get_rank(), dist.get_world_size()

  def broadcast_bytes(self, data, nbytes, root=0):
    import torch
    t = torch.zeros(int(nbytes), dtype=torch.uint8)
    if self.rank == root:
      t = torch.from_numpy(np.frombuffer(bytes(data), dtype=np.uint8).copy())
    self._dist.broadcast(t, root)
    return t.numpy().tobytes()

  def allgather_bytes(self, data):
    import torch
    mine = torch.from_numpy(np.frombuffer(bytes(data), dtype=np.uint8).copy())
    out = [torch.empty_like(mine) for _ in range(self.size)]
    self._dist.all_gather(out, mine)
    return [t.numpy().tobytes() for t in out]

  def allreduce_max(self, value):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
    return float(t.item())
