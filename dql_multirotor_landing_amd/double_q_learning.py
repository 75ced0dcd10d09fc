"""Drop-in mirror of the reference's `double_q_learning.py` (pkg/double_q_learning.py:32-146): same class,
attributes, methods, file names and `.npy` layout.  The table arithmetic (argmax, TD update, transfer scaling) runs
on the device through the C ABI (`dql_agent_predict`, `dql_agent_update`, `dql_agent_transfer`); the host draws the
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
from .mdp import pack_state

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
        self.Q_table_a = np.zeros(shape)
        self.Q_table_b = np.zeros(shape)
        self.state_action_counter = np.zeros(shape)
        self._device = device
        self._lib = None
        self._res = None      # dql_agent*: the tables resident on the device between calls (created on first use)
        self._shadow = None   # what the device holds, padded to 5 levels: (qa, qb, count)
        self._next = None     # (packed state, greedy action) the last update's kernel computed for its next state, on the tables as they are now

    # ---- resident device tables (include/dql.h dql_agent_*) ----
    def _resident(self):
        """The device copy of the tables, brought up to date if the host arrays were written since (they are public attributes: the
        comparison against the shadow of what was last uploaded IS the dirty flag — 3 x 22 KB of memcmp instead of 3 x 22 KB over PCIe)."""
        lib = self._lib
        if lib is None:
            lib = self._lib = _lib.load()
        if self._res is None:
            h = C.c_void_p()
            _lib.check(lib.dql_agent_create(self._device, C.byref(h)))
            self._res = h
            # call arguments and results of the single-transition calls: allocated (and their pointers taken) once
            io = self._io = {"idx": np.zeros(1, np.int32), "act": np.zeros(1, np.uint8), "sa": np.zeros(1, np.int32), "ns": np.zeros(1, np.int32),
                             "alpha": np.zeros(1), "reward": np.zeros(1), "coin": np.zeros(1, np.uint8), "done": np.zeros(1, np.uint8),
                             "q_new": np.zeros(1), "c_new": np.zeros(1)}
            self._ptr = {k: v.ctypes.data_as(C.c_void_p) for k, v in io.items()}
        host = (self.Q_table_a, self.Q_table_b, self.state_action_counter)
        n = self.curriculum_steps
        sh = self._shadow
        if sh is None or not all(a.shape == (n,) + TABLE_SHAPE[1:] and np.array_equal(a, b.reshape(TABLE_SHAPE)[:n]) for a, b in zip(host, sh)):
            sh = self._shadow = self._padded()
            _lib.check(lib.dql_agent_set_tables(self._res, *[a.ctypes.data_as(C.c_void_p) for a in sh]))
            self._next = None  # the host wrote the tables: what the last update predicted for its next state no longer holds
        return lib, self._res

    def close(self):
        if getattr(self, "_res", None):
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

    def _check_state(self, state, n):
        if len(state) != n:
            raise IndexError(f"expected an index tuple of length {n}")
        dims = (self.curriculum_steps, 3, 3, 3, 7, 3)[:n]
        for v, d in zip(state, dims):
            if not -d <= int(v) < d:
                raise IndexError(f"index {v} is out of bounds for axis with size {d}")
        return tuple(int(v) % d for v, d in zip(state, dims))

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
        lib, res = self._resident()
        cell = pack_state(sa[:5]) * 3 + sa[5]
        io, p = self._io, self._ptr
        io["sa"][0] = cell; io["ns"][0] = pack_state(ns); io["alpha"][0] = alpha; io["reward"][0] = reward
        sel_b = False
        if self.mode == "reference":
            _lib.check(lib.dql_agent_update_resident(res, p["sa"], p["ns"], p["alpha"], float(gamma), p["reward"], 1, Q_REFERENCE, None, None, p["q_new"], p["c_new"], p["act"]))
        else:
            sel_b = not u < 0.5
            io["coin"][0] = 1 if sel_b else 0; io["done"][0] = 1 if done else 0
            _lib.check(lib.dql_agent_update_resident(res, p["sa"], p["ns"], p["alpha"], float(gamma), p["reward"], 1, Q_PAPER, p["coin"], p["done"], p["q_new"], p["c_new"], p["act"]))
        # the kernel reports the one cell it changed and its visit counter: patch the host arrays and the shadow of the device copy
        q_new, c_new = float(io["q_new"][0]), float(io["c_new"][0])
        (self.Q_table_b if sel_b else self.Q_table_a)[sa] = q_new
        self.state_action_counter[sa] = c_new
        self._shadow[1 if sel_b else 0][cell] = q_new
        self._shadow[2][cell] = c_new
        self._next = (int(io["ns"][0]), int(io["act"][0]))

    # ---- pkg/double_q_learning.py:110-124 ----
    def guess(self, state: State, exploration_rate: float):
        explore = np.random.uniform(0, 1) < exploration_rate
        return int(np.where(explore, np.random.randint(3), self.predict(state)))  # randint always drawn (B4)

    def predict(self, state: State):
        s = self._check_state(state, 5)
        lib, res = self._resident()   # (re-uploads and forgets self._next if the host arrays were written)
        idx = pack_state(s)
        if self._next is not None and self._next[0] == idx:
            return self._next[1]        # the update kernel already answered this on the current tables
        self._io["idx"][0] = idx
        _lib.check(lib.dql_agent_predict_resident(res, self._ptr["idx"], 1, self._ptr["act"]))
        return int(self._io["act"][0])

    get_action = guess  # name used by BASELINE.json's north_star
