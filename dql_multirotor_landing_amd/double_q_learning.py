"""Drop-in mirror of the reference's `double_q_learning.py` (pkg/double_q_learning.py:32-146): same class,
attributes, methods, file names and `.npy` layout.  The table arithmetic (argmax, TD update, transfer scaling) runs
on the device through the C ABI (`dql_agent_mirror_predict`, `dql_agent_mirror_update` on tables resident on the device and
mirrored from the public host arrays, `dql_agent_transfer`); the host draws the
reference's `np.random` numbers so that a seeded run consumes the global MT19937 stream exactly as the reference
does (B1, B4)."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Tuple, Union

import numpy as np

import ctypes as C

from . import _lib, ops
from .config import MAX_LEVELS, N_CELLS, Q_PAPER, Q_REFERENCE, TABLE_SHAPE

State = Tuple[int, int, int, int, int]
StateAction = Tuple[int, int, int, int, int, int]

ASSETS_PATH = Path(__file__).resolve().parent.parent / "assets"  # the reference resolves this through rospkg (pkg/__init__.py:5-7)


class DoubleQLearningAgent:
    """Agent that learns and makes decisions (two (S,3,3,3,7,3) float64 tables + visit counter)."""

    def __init__(self, curriculum_steps: int = 5, device: int = 0, mode: str = "reference") -> None:
        """`mode` (build-specific): "reference" reproduces the reference's update (B1-B3: always Q_table_a, valued by itself,
        bootstrap only when the position bin changed); "paper" is Double Q-learning as the code was meant to be: the uniform
        draw of `update` picks the table, the other one values its greedy action, and the bootstrap stops at terminal
        transitions (`update(..., done=...)`)."""
        if not 1 <= curriculum_steps <= MAX_LEVELS:
            raise ValueError(f"curriculum_steps must be in 1..{MAX_LEVELS}")
        if mode not in ("reference", "paper"):
            raise ValueError("mode must be 'reference' or 'paper'")
        self.mode = mode
        self.curriculum_steps = curriculum_steps
        shape = (curriculum_steps,) + TABLE_SHAPE[1:]
        self._pending = False  # an update whose cell the library has not patched into the arrays below yet (see `update`)
        self._res = None      # dql_agent*: the tables resident on the device between calls (created on first use)
        self.Q_table_a = np.zeros(shape)
        self.Q_table_b = np.zeros(shape)
        self.state_action_counter = np.zeros(shape)
        self._device = device
        self._lib = None
        self._bound = None    # (Q_table_a, Q_table_b, state_action_counter) objects whose buffer addresses self._ptrs holds
        self._ptrs = None

    # ---- the three public tables (pkg/double_q_learning.py:35-40).  Plain numpy arrays, readable, writable and replaceable at any time, as in the reference;
    # they are properties only so that reading or replacing one first lets the library finish an update it still has in flight (`update` returns when the
    # kernel is launched; the one changed cell and its visit counter are patched into these arrays at the next access through the agent, whatever it is) ----
    def _complete(self):
        if self._pending:
            self._pending = False
            rc = self._lib.dql_agent_mirror_complete(self._res)
            if rc:
                _lib.check(rc)

    Q_table_a = property(lambda self: (self._complete(), self._qa)[1], lambda self, v: (self._complete(), setattr(self, "_qa", v))[0])
    Q_table_b = property(lambda self: (self._complete(), self._qb)[1], lambda self, v: (self._complete(), setattr(self, "_qb", v))[0])
    state_action_counter = property(lambda self: (self._complete(), self._cnt)[1], lambda self, v: (self._complete(), setattr(self, "_cnt", v))[0])

    # ---- resident device tables, mirrored from the public host arrays (include/dql.h dql_agent_mirror_*) ----
    def _resident(self):
        """(lib, agent handle, the three table pointers, level count).  The host arrays are public attributes that callers read, write
        and replace: the library compares them with what the device holds on every call (memcmp against its shadow, ~1 us a table) and
        uploads what differs — the comparison IS the dirty flag.  Here only the buffer addresses are kept, and taken again when an
        attribute was rebound to another array (anything that is not writable C-order float64 of the table's shape is replaced by such
        a copy first, as `np.save` / the device would see it)."""
        lib = self._lib
        if lib is None:
            lib = self._lib = _lib.load()
        if self._res is None:
            h = C.c_void_p()
            _lib.check(lib.dql_agent_create(self._device, C.byref(h)))
            self._res = h
            self._act = C.c_uint8(0)
            self._act_ref = C.byref(self._act)
        a, b, c = self._qa, self._qb, self._cnt  # (the raw attributes: the library call that follows completes a pending update itself)
        bound = self._bound
        if bound is None or a is not bound[0] or b is not bound[1] or c is not bound[2]:
            shape = (self.curriculum_steps,) + TABLE_SHAPE[1:]
            fixed = []
            for t in (a, b, c):
                if not (isinstance(t, np.ndarray) and t.dtype == np.float64 and t.flags.c_contiguous and t.flags.writeable and t.flags.aligned):
                    t = np.array(t, dtype=np.float64, order="C")
                if t.shape != shape:
                    raise ValueError(f"table of shape {t.shape}, expected {shape}")
                fixed.append(t)
            self._complete()  # (before the arrays the pending patch points into are let go)
            a, b, c = self._qa, self._qb, self._cnt = fixed
            self._bound = (a, b, c)
            self._ptrs = tuple(t.ctypes.data for t in fixed)
        return lib, self._res, self._ptrs, self.curriculum_steps

    def close(self):
        if getattr(self, "_res", None):
            self._complete()
            _lib.load().dql_agent_destroy(self._res)
            self._res = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- persistence: pkg/double_q_learning.py:42-75 ----
    def save(self, save_path: Path):
        save_path = Path(save_path)
        for name, arr in (("Q_table_a.npy", self.Q_table_a), ("Q_table_b.npy", self.Q_table_b), ("state_action_count.npy", self.state_action_counter)):
            tmp = save_path / f".{name}.{os.getpid()}.tmp"  # same bytes as the reference's np.save; a reader never sees half a table
            with open(tmp, "wb") as f:
                np.save(f, np.ascontiguousarray(arr, dtype=np.float64))
            os.replace(tmp, save_path / name)

    @staticmethod
    def load(save_path: Path = ASSETS_PATH):
        save_path = Path(save_path)
        with open(save_path / "Q_table_a.npy", "rb") as f:
            qa = np.load(f, allow_pickle=False)
        with open(save_path / "Q_table_b.npy", "rb") as f:
            qb = np.load(f, allow_pickle=False)
        with open(save_path / "state_action_count.npy", "rb") as f:
            sac = np.load(f, allow_pickle=False)
        if qa.shape != qb.shape != sac.shape:  # chained comparison kept as in the reference (:66, B12)
            raise ValueError(f"The shapes of Q table a {qa.shape}, Q table b {qb.shape}and State action count {sac.shape} cannot be different")
        agent = DoubleQLearningAgent(len(qa))
        agent.Q_table_a, agent.Q_table_b, agent.state_action_counter = qa, qb, sac
        return agent

    # ---- device round trips: tables padded to the 5-level device layout ----
    def _padded(self):
        def pad(t):
            out = np.zeros(TABLE_SHAPE, dtype=np.float64)
            out[: self.curriculum_steps] = t
            return out.reshape(-1)
        if self.curriculum_steps == MAX_LEVELS:
            return tuple(np.array(t, dtype=np.float64, order="C").reshape(-1) for t in (self.Q_table_a, self.Q_table_b, self.state_action_counter))
        return pad(self.Q_table_a), pad(self.Q_table_b), pad(self.state_action_counter)

    def _unpad(self, qa, qb, cnt):
        n = self.curriculum_steps
        self.Q_table_a = qa.reshape(TABLE_SHAPE)[:n].copy()
        self.Q_table_b = qb.reshape(TABLE_SHAPE)[:n].copy()
        self.state_action_counter = cnt.reshape(TABLE_SHAPE)[:n].copy()

    _DIMS = (None, 3, 3, 3, 7, 3)

    def _check_state(self, state, n):
        """numpy's index semantics for the tuple (IndexError out of bounds, negative indices wrap), returned as plain ints"""
        if len(state) != n:
            raise IndexError(f"expected an index tuple of length {n}")
        out = []
        d = self.curriculum_steps
        for k in range(n):
            v = int(state[k])
            if not 0 <= v < d:
                if not -d <= v < 0:
                    raise IndexError(f"index {v} is out of bounds for axis with size {d}")
                v += d
            out.append(v)
            d = self._DIMS[k + 1] if k + 1 < 6 else 0
        return tuple(out)

    # ---- pkg/double_q_learning.py:77-89 ----
    def transfer_learning(self, current_curriculum_step: int, transfer_learning_ratio: float):
        k = int(current_curriculum_step)
        if not 0 <= k < self.curriculum_steps:
            raise IndexError(f"index {k} is out of bounds for axis 0 with size {self.curriculum_steps}")
        if self.curriculum_steps != MAX_LEVELS and k == 0:
            # the k = 0 wrap reads the LAST level of the table (B6); with a shorter table do the wrap on the padded copy
            qa, qb, cnt = self._padded()
            src = self.curriculum_steps - 1
            qa.reshape(TABLE_SHAPE)[MAX_LEVELS - 1] = qa.reshape(TABLE_SHAPE)[src]
            qb.reshape(TABLE_SHAPE)[MAX_LEVELS - 1] = qb.reshape(TABLE_SHAPE)[src]
        else:
            qa, qb, cnt = self._padded()
        ops.agent_transfer(qa, qb, k, transfer_learning_ratio, device=self._device)
        self._unpad(qa, qb, cnt)

    # ---- pkg/double_q_learning.py:91-108, 126-146 ----
    def update(self, current_state_action: StateAction, next_state: State, alpha: float, gamma: float, reward, done: bool = False):
        sa = self._check_state(current_state_action, 6)
        ns = self._check_state(next_state, 5)
        u = np.random.uniform(0, 1)  # reference: drawn and ignored, both arms select Q_table_a (B1); paper mode: the coin
        lib, res, (pa, pb, pc), n = self._resident()
        cell = ((((sa[0] * 3 + sa[1]) * 3 + sa[2]) * 3 + sa[3]) * 7 + sa[4]) * 3 + sa[5]
        nidx = (((ns[0] * 3 + ns[1]) * 3 + ns[2]) * 3 + ns[3]) * 7 + ns[4]
        # the library patches the one changed cell and its visit counter into Q_table_a / _b / state_action_counter itself — when the kernel launched here has
        # finished: at the next call on this agent or the next read of a table through it (`_complete`), so the kernel's round trip hides behind whatever the
        # caller does between update() and its next guess() (the reference's loop: logging, the success deque, pkg/trainer.py:204-232)
        if self.mode == "reference":
            rc = lib.dql_agent_mirror_update_deferred(res, pa, pb, pc, n, cell, nidx, float(alpha), float(gamma), float(reward), Q_REFERENCE, 0, 0)
        else:
            rc = lib.dql_agent_mirror_update_deferred(res, pa, pb, pc, n, cell, nidx, float(alpha), float(gamma), float(reward), Q_PAPER, 0 if u < 0.5 else 1, 1 if done else 0)
        if rc:
            _lib.check(rc)
        self._pending = True

    # ---- pkg/double_q_learning.py:110-124 ----
    def guess(self, state: State, exploration_rate: float):
        explore = np.random.uniform(0, 1) < exploration_rate
        rnd = np.random.randint(3)  # always drawn (B4: np.where evaluates both arms)
        greedy = self.predict(state)
        return int(rnd) if explore else greedy

    def predict(self, state: State):
        s = self._check_state(state, 5)
        lib, res, (pa, pb, pc), n = self._resident()
        # answered from the last update's kernel when that update's next state is asked for and the tables were not written since
        rc = lib.dql_agent_mirror_predict(res, pa, pb, pc, n, (((s[0] * 3 + s[1]) * 3 + s[2]) * 3 + s[3]) * 7 + s[4], self._act_ref)
        self._pending = False  # (every mirror call completes a pending update first)
        if rc:
            _lib.check(rc)
        return self._act.value

    get_action = guess  # name used by BASELINE.json's north_star
