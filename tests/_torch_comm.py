"""TEST INFRASTRUCTURE: a torch.distributed (gloo) stand-in for the product's RCCL communicator (comm.RcclComm), so that
the multi-rank logic — env sharding, window exchange schedule, counter sums, episode-log gathers — can run with world
size 2 on a box without GPUs.  Same interface as RcclComm; never imported by the product."""
from __future__ import annotations

import numpy as np


class HostWindowReducer:
    """The window exchange through host memory: get_accum -> all_reduce(sum) -> set_accum."""

    def __init__(self, engine, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.engine = torch, dist, group, engine

    def all_reduce(self):
        t = self.torch.from_numpy(np.ascontiguousarray(self.engine.get_accum()))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self.engine.set_accum(t.numpy())


class TorchComm:
    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def reducer(self, engine):
        return HostWindowReducer(engine, self.group)

    def _reduce(self, v, op):
        t = self.torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
        self.dist.all_reduce(t, op=op, group=self.group)
        return t.numpy()

    def all_reduce_sum(self, v):
        return self._reduce(v, self.dist.ReduceOp.SUM)

    def all_reduce_max(self, v):
        return self._reduce(v, self.dist.ReduceOp.MAX)

    def all_gather_masks(self, done, goal):
        loc = np.stack([np.ascontiguousarray(done, dtype=np.uint64), np.ascontiguousarray(goal, dtype=np.uint64)])
        t = self.torch.from_numpy(loc.view(np.int64))
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        full = np.concatenate([o.numpy().view(np.uint64) for o in out], axis=2)
        return full[0], full[1]

    def barrier(self):
        self.dist.barrier(group=self.group)
