"""`Engine`: one GPU's shard of vectorised landing environments + the Q tables, over the C ABI.

This is the vectorised counterpart of `gym.make("Landing-Training-v0")` + `DoubleQLearningAgent()` in the
reference's trainer (pkg/trainer.py:46-48,176-183): `train_steps` is the loop body of pkg/trainer.py:191-212 for
all envs at once.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .config import ACC_LEN, CHECK_NAMES, DqlConfig, N_CELLS, TABLE_SHAPE


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    def __init__(self, cfg: DqlConfig, n_envs: int, seed: int = 42, device: int = 0, env_id_offset: int = 0, alpha_table=None):
        self.lib = _lib.load()
        self.cfg = cfg
        self.n = int(n_envs)
        self._c = cfg.to_c()
        h = C.c_void_p()
        _lib.check(self.lib.dql_create(C.byref(self._c), device, self.n, seed, env_id_offset, C.byref(h)))
        self._h = h
        tab = cfg.alpha_table() if alpha_table is None else np.ascontiguousarray(alpha_table, dtype=np.float64)
        _lib.check(self.lib.dql_set_alpha_table(self._h, _p(tab), len(tab)))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dql_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stepping ----
    def train_steps(self, n_steps: int, eps: float):
        _lib.check(self.lib.dql_train_steps(self._h, int(n_steps), float(eps)))

    def eval_steps(self, n_steps: int):
        _lib.check(self.lib.dql_eval_steps(self._h, int(n_steps)))

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.uint8)
        if a.shape != (self.n,):
            raise ValueError(f"actions must have shape ({self.n},)")
        if (a > 10).any() or ((a & 3) > 2).any():  # ax | ay << 2 with ax, ay in 0..2: at most 2 | 2 << 2 = 10 (the library checks again)
            raise ValueError("actions must be ax | ay << 2 with ax, ay in 0 (increase), 1 (decrease), 2 (hold)")
        if not self.cfg.two_axis and ((a >> 2) % 2 != 0).any():  # ay in {0 (not given), 2 (hold)}: pkg/mdp.py:544-545 raises for anything else
            raise ValueError("Cannot move in the y direction while training")
        _lib.check(self.lib.dql_step(self._h, _p(a)))

    def step_raw(self, a):
        """`step` without the argument checks: `a` is a contiguous uint8[n] the caller vouches for (the kernel checks the codes again)"""
        _lib.check(self.lib.dql_step(self._h, a.ctypes.data_as(C.c_void_p)))

    def step_dev(self, dev_ptr: int):
        """one agent step with actions that already live in device memory (n uint8, e.g. written by the caller's own policy kernel)"""
        _lib.check(self.lib.dql_step_dev(self._h, C.c_void_p(int(dev_ptr))))

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        _lib.check(self.lib.dql_reset(self._h, _p(m)))

    def set_curriculum(self, level: int):
        _lib.check(self.lib.dql_set_curriculum(self._h, int(level)))
        self.cfg.working_curriculum_step = int(level)

    def sync(self):
        _lib.check(self.lib.dql_sync(self._h))

    # ---- per-env results ----
    def states(self):
        idx = np.zeros(self.n, dtype=np.int32)
        _lib.check(self.lib.dql_get_states(self._h, _p(idx), None))
        return idx

    def rewards(self):
        r = np.zeros(self.n, dtype=np.float64)
        _lib.check(self.lib.dql_get_rewards(self._h, _p(r)))
        return r

    def dones(self):
        d = np.zeros(self.n, dtype=np.uint8); c = np.zeros(self.n, dtype=np.int8)
        _lib.check(self.lib.dql_get_dones(self._h, _p(d), _p(c)))
        return d, c

    def actions(self):
        a = np.zeros(self.n, dtype=np.uint8)
        _lib.check(self.lib.dql_get_actions(self._h, _p(a)))
        return a

    def obs(self):
        o = np.zeros((6, self.n), dtype=np.float64)
        _lib.check(self.lib.dql_get_obs(self._h, _p(o)))
        return o

    def n_fields(self):
        a, b = C.c_int32(), C.c_int32()
        _lib.check(self.lib.dql_n_fields(C.byref(a), C.byref(b)))
        return a.value, b.value

    def field_names(self, is_int=False):
        cache = self.__dict__.setdefault("_names", {})
        if is_int not in cache:  # 64 + 7 ctypes calls: made once (the single-env drop-ins ask every step)
            n = self.n_fields()[1 if is_int else 0]
            cache[is_int] = [self.lib.dql_field_name(i, int(is_int)).decode() for i in range(n)]
        return list(cache[is_int])

    def step_outputs(self):
        """what `TrainingLandingEnv.step` returns, for every env, in one device round trip (include/dql.h dql_step_outputs):
        dict of arrays idx_x, idx_y, reward, done, code, step_count, cumulative_reward, was_reset (fresh copies of the engine's buffers)"""
        so = self.__dict__.get("_so")
        if so is None:  # output arrays and their pointers: made once
            n = self.n
            o = {"idx_x": np.zeros(n, np.int32), "idx_y": np.zeros(n, np.int32), "reward": np.zeros(n, np.float64), "done": np.zeros(n, np.uint8),
                 "code": np.zeros(n, np.int8), "step_count": np.zeros(n, np.int32), "cumulative_reward": np.zeros(n, np.float64), "was_reset": np.zeros(n, np.uint8)}
            so = self._so = (o, [_p(o[k]) for k in ("idx_x", "idx_y", "reward", "done", "code", "step_count", "cumulative_reward", "was_reset")])
        _lib.check(self.lib.dql_step_outputs(self._h, *so[1]))
        return {k: v.copy() for k, v in so[0].items()}

    def step_outputs_view(self):
        """the same without the copies: the engine's own buffers, overwritten by the next call"""
        if self.__dict__.get("_so") is None:
            self.step_outputs()
            return self._so[0]
        _lib.check(self.lib.dql_step_outputs(self._h, *self._so[1]))
        return self._so[0]

    def get_fields(self):
        nr, ni = self.n_fields()
        reals = np.zeros((nr, self.n), dtype=np.float64); ints = np.zeros((ni, self.n), dtype=np.int32)
        _lib.check(self.lib.dql_get_sim_state(self._h, _p(reals), nr))
        _lib.check(self.lib.dql_get_sim_ints(self._h, _p(ints), ni))
        return reals, ints

    def set_fields(self, reals, ints):
        reals = np.ascontiguousarray(reals, dtype=np.float64); ints = np.ascontiguousarray(ints, dtype=np.int32)
        _lib.check(self.lib.dql_set_sim_state(self._h, _p(reals), reals.shape[0]))
        _lib.check(self.lib.dql_set_sim_ints(self._h, _p(ints), ints.shape[0]))

    # ---- tables ----
    def get_tables(self):
        qa = np.zeros(N_CELLS); qb = np.zeros(N_CELLS); cnt = np.zeros(N_CELLS)
        _lib.check(self.lib.dql_get_tables(self._h, _p(qa), _p(qb), _p(cnt)))
        return qa.reshape(TABLE_SHAPE), qb.reshape(TABLE_SHAPE), cnt.reshape(TABLE_SHAPE)

    def get_counts(self):
        """the visit counter alone (what the Trainer's per-chunk learning-rate report needs): the same call as get_tables, one table over the bus"""
        cnt = np.zeros(N_CELLS)
        _lib.check(self.lib.dql_get_tables(self._h, None, None, _p(cnt)))
        return cnt.reshape(TABLE_SHAPE)

    def set_tables(self, qa=None, qb=None, count=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64).ravel()
        qa, qb, count = f(qa), f(qb), f(count)
        for a in (qa, qb, count):
            if a is not None and a.size != N_CELLS:
                raise ValueError(f"tables must have shape {TABLE_SHAPE}")
        _lib.check(self.lib.dql_set_tables(self._h, _p(qa), _p(qb), _p(count)))

    def transfer(self, k: int, ratio: float):
        _lib.check(self.lib.dql_transfer(self._h, int(k), float(ratio)))

    # ---- multi-GPU exchange ----
    def set_windowed(self, on: bool):
        _lib.check(self.lib.dql_set_windowed(self._h, int(on)))

    def accum_dev_ptr(self):
        p = C.c_void_p(); n = C.c_int64()
        _lib.check(self.lib.dql_diag_accum_dev_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_window_buffer(self, dev_ptr):
        _lib.check(self.lib.dql_set_window_buffer(self._h, C.c_void_p(dev_ptr) if dev_ptr else None))

    def stream_handle(self):
        p = C.c_void_p()
        _lib.check(self.lib.dql_stream_handle(self._h, C.byref(p)))
        return p.value

    def flush(self):
        _lib.check(self.lib.dql_flush(self._h))

    def apply_accum(self):
        _lib.check(self.lib.dql_apply_accum(self._h))

    def get_accum(self):
        a = np.zeros(ACC_LEN, dtype=np.int64)
        _lib.check(self.lib.dql_get_accum(self._h, _p(a)))
        return a

    def set_accum(self, a):
        a = np.ascontiguousarray(a, dtype=np.int64)
        _lib.check(self.lib.dql_set_accum(self._h, _p(a)))

    def attach_comm(self, comm_handle):
        """comm_handle: the dql_comm* of comm.RcclComm (None detaches)"""
        _lib.check(self.lib.dql_attach_comm(self._h, comm_handle))

    def allreduce_window(self):
        _lib.check(self.lib.dql_allreduce_window(self._h))

    # ---- one-shot peer-to-peer exchange (include/dql.h dql_p2p_*) ----
    def p2p_create(self, rank: int, world: int) -> bytes:
        """allocates this rank's exchange buffer; returns its 64-byte HIP IPC handle (to be gathered from all ranks)"""
        h = (C.c_uint8 * _lib.P2P_HANDLE_BYTES)()
        _lib.check(self.lib.dql_p2p_create(self._h, int(rank), int(world), h))
        return bytes(h)

    def p2p_connect(self, handles):
        """handles: every rank's handle, in rank order"""
        blob = b"".join(handles)
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _lib.check(self.lib.dql_p2p_connect(self._h, buf))

    def p2p_connect_local(self, peers):
        """peers: the Engines of ALL ranks in rank order, None where a rank lives in another process (those: p2p_connect)"""
        arr = (C.c_void_p * len(peers))(*[None if e is None else e._h for e in peers])
        _lib.check(self.lib.dql_p2p_connect_local(self._h, arr))

    def p2p_exchange_window(self):
        _lib.check(self.lib.dql_p2p_exchange_window(self._h))

    def p2p_push_window(self):
        _lib.check(self.lib.dql_p2p_push_window(self._h))

    def p2p_wait_window(self):
        _lib.check(self.lib.dql_p2p_wait_window(self._h))

    def p2p_failed_seq(self) -> int:
        """sequence number (1-based) of the first exchange that gave up on a missing peer, 0 = none.  `stats()` raises RuntimeError once
        this is non-zero: the training loop's per-chunk synchronisation point ends a run whose replicas have diverged."""
        v = C.c_int32(0)
        _lib.check(self.lib.dql_p2p_status(self._h, C.byref(v)))
        return int(v.value)

    def p2p_failed(self) -> bool:
        return self.p2p_failed_seq() != 0

    def sync_time_ms(self):
        ms = C.c_double(); n = C.c_int64()
        _lib.check(self.lib.dql_diag_sync_time_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- checkpoint / resume ----
    def step_index(self) -> int:
        v = C.c_int64()
        _lib.check(self.lib.dql_get_step_index(self._h, C.byref(v)))
        return v.value

    def set_step_index(self, j: int):
        _lib.check(self.lib.dql_set_step_index(self._h, int(j)))

    def publish_tables(self):
        _lib.check(self.lib.dql_publish_tables(self._h))

    # ---- stats / timing / knobs ----
    def stats(self):
        s = _lib.DqlStatsC()
        _lib.check(self.lib.dql_stats_get(self._h, C.byref(s)))
        return {"agent_steps": s.agent_steps, "decisions": s.decisions, "episodes": s.episodes,
                "by_code": {CHECK_NAMES[i]: s.by_code[i] for i in range(len(CHECK_NAMES))},
                "reward_sum": s.reward_sum, "physics_ticks": s.physics_ticks}

    def stats_reset(self):
        _lib.check(self.lib.dql_stats_reset(self._h))

    def timer_start(self):
        _lib.check(self.lib.dql_diag_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_double()
        _lib.check(self.lib.dql_diag_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def kernel_timer(self, on: bool):
        _lib.check(self.lib.dql_diag_kernel_timer(self._h, int(on)))

    def kernel_time_ms(self):
        ms = C.c_double(); n = C.c_int64()
        _lib.check(self.lib.dql_diag_kernel_time_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def delay(self, microseconds: float):
        _lib.check(self.lib.dql_diag_delay(self._h, float(microseconds)))

    def set_option(self, name: str, value: int):
        _lib.check(self.lib.dql_set_option(self._h, name.encode(), int(value)))

    # ---- episode log (completion order for the promotion rule, pkg/trainer.py:218-232) ----
    def episode_log_enable(self, capacity_periods: int):
        _lib.check(self.lib.dql_episode_log_enable(self._h, int(capacity_periods)))
        self._elog_cap = int(capacity_periods)

    def episode_log_read(self, words=None):
        """(done, goal) uint64[n_periods, n_waves]: bit l of word w = env 64 w + l finished an episode / finished it in the goal state.
        words = k: only the first k words of every period (the first 64 k envs) cross the bus — what a caller judging a few envs needs."""
        nw = (self.n + 63) // 64
        if words is not None:
            nw = max(0, min(int(words), nw))
        done = np.zeros((self._elog_cap, nw), dtype=np.uint64)
        goal = np.zeros((self._elog_cap, nw), dtype=np.uint64)
        k = C.c_int32()
        if words is None:
            _lib.check(self.lib.dql_episode_log_read(self._h, _p(done), _p(goal), self._elog_cap, C.byref(k)))
        else:
            _lib.check(self.lib.dql_episode_log_read_words(self._h, _p(done), _p(goal), self._elog_cap, nw, C.byref(k)))
        return done[:k.value], goal[:k.value]

    def state_bytes_per_env(self) -> int:
        b = C.c_int64()
        _lib.check(self.lib.dql_state_bytes_per_env(self._h, C.byref(b)))
        return b.value
