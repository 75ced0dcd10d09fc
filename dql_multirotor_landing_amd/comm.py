"""`RcclComm`: one rank of a one-process-per-GPU job, over the C ABI's RCCL binding (`dql_comm_*`, include/dql.h).

No PyTorch: `libdql_hip.so` loads librccl itself.  The only thing ranks need out of band is rank 0's 128-byte
`ncclUniqueId`; it travels through a small file that every rank of the job can name without talking to the others:

    $DQL_COMM_ID_FILE                                     if set (the launcher chooses), else
    $TMPDIR/dql_comm_<uid>_<ppid>_<parent start>_<MASTER_PORT>_<n>.id

All ranks of a node share their parent (torch.distributed.run's agent, or `bench.py --gpus N` itself), and (pid, start
time) of a live process is unique, so a file left behind by an earlier job can never be mistaken for this one's; `<n>`
counts the communicators a process has created (every rank creates them in the same order).  Rank 0 writes the file
atomically (temp + rename) and removes it when the communicator is closed.  An explicit $DQL_COMM_ID_FILE has no such nonce
in its name, so every bootstrap file also carries one INSIDE: a 24-byte header (magic, MASTER_PORT, WORLD_SIZE, 12 bytes of job nonce).
The nonce is a hash of $DQL_COMM_JOB_ID when the launcher sets one (ranks that do not share a parent: mpirun, srun), else of
$TORCHELASTIC_RUN_ID + $TORCHELASTIC_RESTART_COUNT + the parent's (pid, start time) — the same for every rank of a job, different for any other
job and for any other ATTEMPT of an elastic job, whenever it runs.  A parent that respawns its ranks itself on the same port must hand every
spawn a fresh $DQL_COMM_JOB_ID (bench.py's spawn_ranks does: a uuid4 per spawn).
A reader accepts a file with its own job's header only: what a killed job left behind — even one relaunched seconds later on the same
port — is polled past, however old or young it is, and a rank that starts minutes after rank 0 still finds rank 0's file valid.
Rank 0 removes whatever sits under the name before it creates its communicator.

Launch contract (same variables torch.distributed.run exports): RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import tempfile
import time
from pathlib import Path
from typing import Optional

import numpy as np

from . import _lib

_created = 0          # communicators created by this process so far
_env_comm = None      # the job's communicator (from_env), shared by everything in this process


def _job_nonce() -> bytes:
    """12 bytes every rank of THIS job computes alike and no other job does (module docstring)"""
    import hashlib
    job = os.environ.get("DQL_COMM_JOB_ID")
    # TORCHELASTIC_RESTART_COUNT: an elastic restart keeps the agent (the parent), the run id and — with static rendezvous — the port; without it the
    # workers of attempt k + 1 would compute attempt k's nonce and could pick up the dead attempt's unique id before rank 0 has replaced it
    key = (f"job:{job}" if job else
           f"run:{os.environ.get('TORCHELASTIC_RUN_ID', '')}|restart:{os.environ.get('TORCHELASTIC_RESTART_COUNT', '')}|parent:{_parent_token()}")
    return hashlib.sha256(key.encode()).digest()[:12]


def _header(world: int) -> bytes:
    port = int(os.environ.get("MASTER_PORT", "0") or 0) & 0xFFFFFFFF
    return b"DQLD" + port.to_bytes(4, "little") + int(world).to_bytes(4, "little") + _job_nonce()


def _read_fresh(f: Path, world: int, n_payload: int) -> Optional[bytes]:
    """payload of bootstrap file `f` if it is this job's (header with the job nonce), else None"""
    try:
        b = f.read_bytes()
    except OSError:
        return None
    h = _header(world)
    if len(b) != len(h) + n_payload or b[:len(h)] != h:
        return None
    return b[len(h):]


def _parent_token() -> str:
    ppid = os.getppid()
    try:  # field 22 of /proc/<pid>/stat: start time in clock ticks since boot (the command name may contain spaces: split after it)
        start = Path(f"/proc/{ppid}/stat").read_text().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        start = "0"
    return f"{ppid}_{start}"


def id_file_path(seq: int) -> Path:
    explicit = os.environ.get("DQL_COMM_ID_FILE")
    if explicit:
        return Path(explicit if seq == 0 else f"{explicit}.{seq}")
    port = os.environ.get("MASTER_PORT", "0")
    return Path(tempfile.gettempdir()) / f"dql_comm_{os.getuid()}_{_parent_token()}_{port}_{seq}.id"


def gather_via_files(tag: str, rank: int, world: int, payload: bytes, timeout_s: float = 600.0):
    """All-gather of one small byte string per rank through files every sibling rank can name (same naming scheme as the
    communicator's unique id): for bootstrap data such as the HIP IPC handles of the peer-to-peer exchange, which must not depend
    on a collective library.  Returns the payloads in rank order; every rank removes its own file once all are read."""
    base = id_file_path(0)
    mine = base.with_name(f"{base.stem}.{tag}.{rank}.bin")
    done = base.with_name(f"{base.stem}.{tag}.{rank}.done")
    try:  # a marker of an earlier job under the same explicit name must not tell a peer that THIS rank has read everything
        done.unlink()
    except OSError:
        pass
    tmp = mine.with_suffix(f".{os.getpid()}.tmp")
    tmp.write_bytes(_header(world) + payload)
    os.replace(tmp, mine)
    out, t0 = [], time.monotonic()
    for r in range(world):
        f = base.with_name(f"{base.stem}.{tag}.{r}.bin")
        while True:
            b = _read_fresh(f, world, len(payload))
            if b is not None:
                break
            if time.monotonic() - t0 > timeout_s:
                raise RuntimeError(f"rank {rank}: nothing from rank {r} after {timeout_s:.0f} s ({f})")
            time.sleep(0.01)
        out.append(b)
    # a second round of marker files: nobody deletes what a slower rank has not read yet
    done.write_bytes(_header(world))
    for r in range(world):
        d = base.with_name(f"{base.stem}.{tag}.{r}.done")
        while _read_fresh(d, world, 0) is None:
            if time.monotonic() - t0 > timeout_s:
                raise RuntimeError(f"rank {rank}: rank {r} never finished reading ({d})")
            time.sleep(0.01)
    if rank == 0:  # all ranks have read everything and written their marker: rank 0 clears the data files, each rank its own marker later
        for r in range(world):
            try:
                base.with_name(f"{base.stem}.{tag}.{r}.bin").unlink()
            except OSError:
                pass
    atexit.register(lambda: done.exists() and done.unlink())
    return out


class RcclComm:
    """A rank's RCCL communicator + the Trainer's control-plane exchanges (per-chunk counters: sum; bench timing: max;
    judged envs' episode logs: gather in rank order = global env order)."""

    def __init__(self, rank: int, world: int, device: int = 0, timeout_s: float = 600.0):
        global _created
        self.lib = _lib.load()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        if not 0 <= self.rank < self.world:
            raise ValueError("rank must be in 0 .. world-1")
        n = C.c_int(0)
        _lib.check(self.lib.dql_device_count(C.byref(n)))
        if self.device >= n.value:
            raise RuntimeError(f"rank {self.rank} of {self.world} needs GPU {self.device}, but only {n.value} GPU(s) are visible: one process per GPU, no sharing")
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # single node: bootstrap over loopback (the hostname may not resolve)
        seq, _created = _created, _created + 1
        self._id_file = id_file_path(seq)
        uid = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        if self.rank == 0:
            try:  # whatever an earlier job left under this name goes before this job's id exists
                self._id_file.unlink()
            except OSError:
                pass
            _lib.check(self.lib.dql_comm_unique_id(uid))
            if self.world > 1:
                tmp = self._id_file.with_suffix(f".{os.getpid()}.tmp")
                tmp.write_bytes(_header(self.world) + bytes(uid))
                os.replace(tmp, self._id_file)
        else:
            t0 = time.monotonic()
            while True:
                b = _read_fresh(self._id_file, self.world, _lib.COMM_ID_BYTES)
                if b is not None:
                    break
                if time.monotonic() - t0 > timeout_s:
                    raise RuntimeError(f"rank {self.rank}: no unique id from rank 0 after {timeout_s:.0f} s ({self._id_file})")
                time.sleep(0.02)
            uid = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(b)
        h = C.c_void_p()
        _lib.check(self.lib.dql_comm_create(self.device, self.rank, self.world, uid, C.byref(h)))
        self.handle = h
        atexit.register(self.close)

    @staticmethod
    def from_env(device: Optional[int] = None) -> Optional["RcclComm"]:
        """The job's communicator when this process is one rank of a multi-rank launch (WORLD_SIZE > 1), else None.
        Created once per process."""
        global _env_comm
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world < 2:
            return None
        if _env_comm is None:
            rank = int(os.environ["RANK"])
            dev = int(os.environ.get("LOCAL_RANK", rank)) if device is None else int(device)
            _env_comm = RcclComm(rank, world, dev)
        return _env_comm

    def close(self):
        global _env_comm
        if getattr(self, "handle", None):
            self.lib.dql_comm_destroy(self.handle)
            self.handle = None
            if self.rank == 0 and self.world > 1:
                try:
                    self._id_file.unlink()
                except OSError:
                    pass
        if _env_comm is self:
            _env_comm = None

    # ---- table exchange ----
    def reducer(self, engine):
        from .dist import RcclWindowReducer
        return RcclWindowReducer(engine, self)

    # ---- control plane (host arrays) ----
    def _reduce(self, v, op):
        a = np.ascontiguousarray(v, dtype=np.float64).copy()
        _lib.check(self.lib.dql_comm_allreduce_f64(self.handle, a.ctypes.data_as(C.c_void_p), a.size, op))
        return a

    def all_reduce_sum(self, v) -> np.ndarray:
        return self._reduce(v, _lib.OP_SUM)

    def all_reduce_max(self, v) -> np.ndarray:
        return self._reduce(v, _lib.OP_MAX)

    def all_reduce_sum_i64(self, v) -> np.ndarray:
        a = np.ascontiguousarray(v, dtype=np.int64).copy()
        _lib.check(self.lib.dql_comm_allreduce_i64(self.handle, a.ctypes.data_as(C.c_void_p), a.size, _lib.OP_SUM))
        return a

    def all_gather_masks(self, done: np.ndarray, goal: np.ndarray):
        """[P, W] uint64 on every rank (same shape) -> [P, world * W], rank order"""
        loc = np.ascontiguousarray(np.stack([np.asarray(done, dtype=np.uint64), np.asarray(goal, dtype=np.uint64)]))
        out = np.zeros((self.world,) + loc.shape, dtype=np.uint64)
        _lib.check(self.lib.dql_comm_allgather_u64(self.handle, loc.ctypes.data_as(C.c_void_p), loc.size, out.ctypes.data_as(C.c_void_p)))
        full = np.concatenate(list(out), axis=2)
        return full[0], full[1]

    def barrier(self):
        _lib.check(self.lib.dql_comm_barrier(self.handle))
