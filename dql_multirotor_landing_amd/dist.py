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


# ---- measuring and checking the exchange (bench.py; exercised at world size 2 on CPU by tests/test_dist_gloo.py) ----
def tables_fingerprint(qa, qb, count):
    """6 x 32 bits of SHA-256 over the three tables' bytes, as floats (every 32-bit integer is an exact float64): what ranks compare"""
    import hashlib
    import numpy as np
    h = hashlib.sha256(np.ascontiguousarray(qa, dtype=np.float64).tobytes() + np.ascontiguousarray(qb, dtype=np.float64).tobytes()
                       + np.ascontiguousarray(count, dtype=np.float64).tobytes()).digest()
    return [float(int.from_bytes(h[4 * i:4 * i + 4], "little")) for i in range(6)]


def replicas_identical(engine, comm) -> bool:
    """After an exchange every rank must hold bit-identical tables: all-reduce(max) and all-reduce(min) of the tables' fingerprint agree.
    (The one N > 1 correctness check a run can make about itself; `comm` = RcclComm or any object with all_reduce_max.)"""
    import numpy as np
    fp = tables_fingerprint(*engine.get_tables())
    if comm is None or getattr(comm, "world", 1) == 1:
        return True
    hi = np.asarray(comm.all_reduce_max(fp)); lo = -np.asarray(comm.all_reduce_max([-x for x in fp]))
    return bool((hi == lo).all())


def timed_region(engine, comm, reducer, sync_period: int, steps: int, warmup: int, eps: float, reps: int = 1, final_exchange: bool = True, log=None):
    """`warmup` untimed + `reps` x exactly `steps` timed agent periods on `engine` with table exchanges through `reducer` every `sync_period`
    periods.  Every repetition is bracketed by barrier + device sync on both sides; per repetition the wall time is the MAX over ranks and the
    env-steps the SUM.  A repetition ends on exchanged tables (a window still open after `steps` periods is exchanged inside the clock) unless
    final_exchange is False (the no-exchange yardstick).  A rank on which the device part raises (a peer-to-peer exchange that gave up on a peer)
    STILL takes part in every collective, and every rank then returns None — a failed leg never strands the others in an all-reduce.
    Returns [(wall_s, env_steps, device_ms)] per repetition."""
    import time
    runner = ShardedRunner(engine, reducer, sync_period=sync_period)
    has_timer = hasattr(engine, "timer_start")
    multi = comm is not None and getattr(comm, "world", 1) > 1

    def barrier():
        engine.sync() if hasattr(engine, "sync") else None
        if comm is not None:
            comm.barrier()
            engine.sync() if hasattr(engine, "sync") else None

    out, failed = [], 0.0
    try:
        runner.train_steps(warmup, eps)
        runner.sync()
    except Exception as e:  # noqa: BLE001
        failed = 1.0
        if log:
            log(f"{type(e).__name__}: {e}")
    for _ in range(reps):
        barrier()
        wall, dev_ms, dec = 0.0, 0.0, 0
        if not failed:
            try:
                d0 = engine.stats()["decisions"]
                if has_timer:
                    engine.timer_start()
                t0 = time.perf_counter()
                runner.train_steps(steps, eps)
                if final_exchange:
                    runner.sync()
                if has_timer:
                    dev_ms = engine.timer_stop()       # waits for the stream: this rank's K steps are done
                wall = time.perf_counter() - t0        # (MAX over ranks below = the job's time)
                runner.sync()
                engine.sync() if hasattr(engine, "sync") else None
                dec = engine.stats()["decisions"] - d0
            except Exception as e:  # noqa: BLE001
                failed = 1.0
                if log:
                    log(f"{type(e).__name__}: {e}")
        barrier()
        if multi:
            wall, failed = (float(x) for x in comm.all_reduce_max([wall, failed]))
            dec = int(comm.all_reduce_sum([float(dec)])[0])
        out.append((wall, dec, dev_ms))
    return None if failed else out


def median_repetition(reps, steps: int):
    """(wall_s, env_steps, device_ms) of the repetition with the median throughput, and the spread over the repetitions"""
    order = sorted(range(len(reps)), key=lambda i: reps[i][1] / reps[i][0])
    m = reps[order[len(order) // 2]]
    vals = [r[1] / r[0] for r in reps]
    return m, {"n": len(reps), "value_min": min(vals), "value_max": max(vals), "ms_per_step_min": min(r[0] for r in reps) * 1e3 / steps,
               "ms_per_step_max": max(r[0] for r in reps) * 1e3 / steps, "statistic": "median over back-to-back repetitions of the K-step timed region"}


def compare_exchanges(engine, comm, reducers: dict, primary: str, sync_period: int, steps: int, warmup: int, eps: float, reps: int = 3, log=None, skip_reasons=None):
    """Times every exchange in `reducers` (name -> reducer) on the SAME engine and schedule, in one run, and checks after each leg that the replicas are
    identical.  Returns {name: {"value", "ms_per_step", "replicas_identical", ...} | {"skipped": reason}}; the primary one first (its figures are the run's).
    A leg that fails is reported as skipped, the others still run — but nothing runs after a failed leg on the same engine (its tables may have diverged)."""
    out = {}
    order = [primary] + [k for k in reducers if k != primary]
    broken = None
    for name in order:
        if name not in reducers:
            out[name] = {"skipped": (skip_reasons or {}).get(name, "not set up")}
            continue
        if broken:
            out[name] = {"skipped": f"not run: the {broken} leg failed before it on this engine"}
            continue
        r = timed_region(engine, comm, reducers[name], sync_period, steps, warmup if name != primary else 0, eps, reps=reps, log=log)
        if r is None:
            out[name] = {"skipped": "the exchange failed inside its timed leg (see stderr)"}
            broken = name
            continue
        (w, d, _), spread = median_repetition(r, steps)
        out[name] = {"value": d / w, "ms_per_step": w * 1e3 / steps, "replicas_identical": replicas_identical(engine, comm), "value_min": spread["value_min"], "value_max": spread["value_max"]}
    return out
