"""Multi-GPU exchange: envs shard across ranks with no data-path collective; the only exchange is
the periodic sum of the int64 window accumulators [4][2835] = {fixed-point TD-target sums, visit counts} of the two tables, after
which every rank folds the same totals into its base tables (bit-identical replicas, independent of the order of
summation).  One process per GPU; the sum is `ncclAllReduce(ncclInt64, ncclSum)` issued by `libdql_hip.so` itself on the
engine's HIP stream (`dql_allreduce_window`: RCCL over xGMI, no PyTorch anywhere on this path; comm.py sets the
communicator up).

`ShardedRunner` is backend-agnostic: the engine is anything with train_steps / apply_accum / window access (the HIP
`Engine` in production; CPU tests inject a stand-in engine and a host-side reducer to exercise sharding + reduction with
world size 2)."""
from __future__ import annotations

from typing import Optional


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous env-id range of `rank`; the ids key the per-env RNG, so results do not depend on `world`."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RcclWindowReducer:
    """All-reduce(sum) of the engine's window accumulators: RCCL, in place in device memory, on the engine's stream."""

    def __init__(self, engine, comm):
        self.engine, self.comm = engine, comm
        engine.attach_comm(comm.handle)

    def all_reduce(self):
        self.engine.allreduce_window()  # flushes the last launch into the window first; asynchronous


class P2PWindowReducer:
    """The same sum without a collective (second step): every rank writes its window into a slot of every
    rank's exchange buffer over the direct links and adds the slots up locally (include/dql.h dql_p2p_*).  The HIP IPC handles
    travel once, at construction, through `gather(tag, rank, world, payload) -> [payload of rank 0, 1, ...]` (default:
    comm.gather_via_files).  Opt-in: functionally tested with several ranks on one GPU; its latency against RCCL's all-reduce
    needs a multi-GPU box."""

    _count = 0

    def __init__(self, engine, rank: int, world: int, gather=None):
        from .comm import gather_via_files
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        tag = f"p2p{P2PWindowReducer._count}"
        P2PWindowReducer._count += 1
        handle = engine.p2p_create(self.rank, self.world)
        handles = (gather or gather_via_files)(tag, self.rank, self.world, handle)
        engine.p2p_connect(handles)

    @classmethod
    def local_group(cls, engines):
        """Reducers for ranks that all live in THIS process (one host thread driving several contexts / GPUs): buffers connected by
        pointer, no IPC, nothing to gather.  Drive them with `ShardedGroup`, which enqueues every rank's push before any rank's wait
        (streams of one process may share a hardware queue: a wait kernel queued ahead of the push it waits for would sit there until
        its poll bound)."""
        reds = []
        for r, e in enumerate(engines):
            red = cls.__new__(cls)
            red.engine, red.rank, red.world = e, r, len(engines)
            e.p2p_create(r, len(engines))
            reds.append(red)
        for e in engines:
            e.p2p_connect_local(engines)
        return reds

    def all_reduce(self):
        self.engine.p2p_exchange_window()  # flush + push + signal + wait + sum, asynchronous on the engine's stream


class LocalWindowReducer:
    """World size 1 on the windowed schedule: the sum over one rank is the identity, only the pending launch has to enter the
    window.  Lets a single-GPU run follow exactly the table schedule of a multi-GPU run with the same sync period."""

    def __init__(self, engine, comm=None):
        self.engine = engine

    def all_reduce(self):
        self.engine.flush()


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
        if self.reducer is not None and self._since > 0:
            self.reducer.all_reduce()
            self.engine.apply_accum()
        self._since = 0



class ShardedGroup:
    """Several ranks driven by ONE host thread ("one host thread drives all contexts, or one per GPU"): the same
    schedule as one `ShardedRunner` per rank, with the peer-to-peer exchange issued in two sweeps — all pushes, then all waits + folds."""

    def __init__(self, engines, sync_period: int = 1):
        if sync_period < 1:
            raise ValueError("sync_period must be >= 1")
        self.engines, self.sync_period = list(engines), int(sync_period)
        self.reducers = P2PWindowReducer.local_group(self.engines)
        self._since = 0
        for e in self.engines:
            e.set_windowed(True)

    def train_steps(self, n_steps: int, eps: float):
        left = int(n_steps)
        while left > 0:
            k = min(left, self.sync_period - self._since)
            for e in self.engines:
                e.train_steps(k, eps)
            self._since += k
            left -= k
            if self._since == self.sync_period:
                self.sync()

    def sync(self):
        if self._since > 0:
            for e in self.engines:
                e.p2p_push_window()
            for e in self.engines:
                e.p2p_wait_window()
                e.apply_accum()
        self._since = 0
