"""Multi-GPU exchange: envs shard across ranks with no data-path collective; the only exchange is
the periodic sum of the int64 window accumulators [4][2835] = {fixed-point TD-target sums, visit counts} of the two tables, after
which every rank folds the same totals into its base tables (bit-identical replicas, independent of the order of
summation).  One process per GPU; `torch.distributed` (backend "nccl" = RCCL over xGMI) is plumbing only: it
owns the 90 KB exchange buffer and runs the all-reduce on the engine's HIP stream.

`ShardedRunner` is backend-agnostic: the engine is anything with train_steps / apply_accum / window access (the HIP
`Engine` in production; CPU tests inject a stand-in to exercise sharding + reduction with the gloo backend)."""
from __future__ import annotations

from typing import Optional

import numpy as np

from .config import ACC_LEN


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous env-id range of `rank`; the ids key the per-env RNG, so results do not depend on `world`."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class TorchWindowReducer:
    """All-reduce(sum) of the engine's window accumulators with torch.distributed on the engine's stream."""

    def __init__(self, engine, device_index: int, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.engine = engine
        dev = torch.device("cuda", device_index)
        self.buf = torch.zeros(ACC_LEN, dtype=torch.int64, device=dev)
        engine.set_window_buffer(self.buf.data_ptr())
        self.stream = torch.cuda.ExternalStream(engine.stream_handle(), device=dev)

    def all_reduce(self):
        self.engine.flush()  # the last launch's accumulators enter the window here
        with self.torch.cuda.stream(self.stream):
            self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.SUM, group=self.group)


class HostWindowReducer:
    """Same exchange through host memory (gloo): used by the CPU tests and as a debugging aid."""

    def __init__(self, engine, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.engine = torch, dist, group, engine

    def all_reduce(self):
        t = self.torch.from_numpy(np.ascontiguousarray(self.engine.get_accum()))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self.engine.set_accum(t.numpy())


class ShardedRunner:
    """Runs `n_steps` agent periods on this rank's shard and synchronises the tables every `sync_period` periods."""

    def __init__(self, engine, reducer: Optional[object], sync_period: int = 1):
        if sync_period < 1:
            raise ValueError("sync_period must be >= 1")
        self.engine, self.reducer, self.sync_period = engine, reducer, int(sync_period)
        self._since = 0
        if reducer is not None:
            engine.set_windowed(True)

    def train_steps(self, n_steps: int, eps: float):
        if self.reducer is None:
            self.engine.train_steps(n_steps, eps)
            return
        left = int(n_steps)
        while left > 0:
            k = min(left, self.sync_period - self._since)
            self.engine.train_steps(k, eps)
            self._since += k
            left -= k
            if self._since == self.sync_period:
                self.sync()

    def sync(self):
        if self.reducer is not None:
            self.reducer.all_reduce()
            self.engine.apply_accum()
        self._since = 0


class TorchComm:
    """The Trainer's control-plane exchanges: per-chunk counters (sum) and episode logs (gather in rank order = global env
    order).  Tensors live on the GPU for the nccl (= RCCL) backend, on the host for gloo."""

    def __init__(self, group=None, device_index: int = 0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device_index = device_index
        self.dev = torch.device("cuda", device_index) if self.backend == "nccl" else torch.device("cpu")

    @staticmethod
    def from_env(device_index: int = 0):
        """A communicator when this process is one rank of an initialised torch.distributed job with more than one rank."""
        import sys
        td = sys.modules.get("torch.distributed")  # never imports torch by itself: a single-GPU run does not need it
        if td is None or not td.is_available() or not td.is_initialized() or td.get_world_size() < 2:
            return None
        return TorchComm(None, device_index)

    def reducer(self, engine):
        if self.backend == "nccl":
            return TorchWindowReducer(engine, self.device_index, self.group)
        return HostWindowReducer(engine, self.group)

    def all_reduce_sum(self, v: np.ndarray) -> np.ndarray:
        t = self.torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def all_gather_masks(self, done: np.ndarray, goal: np.ndarray):
        """[P, W] uint64 on every rank (same shape) -> [P, world * W], rank order"""
        loc = np.stack([np.ascontiguousarray(done, dtype=np.uint64), np.ascontiguousarray(goal, dtype=np.uint64)])
        t = self.torch.from_numpy(loc.view(np.int64)).to(self.dev)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        full = np.concatenate([o.cpu().numpy().view(np.uint64) for o in out], axis=2)
        return full[0], full[1]
