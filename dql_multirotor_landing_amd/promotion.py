"""The reference's promotion rule (pkg/trainer.py:218-232) evaluated over a vectorised run.

Reference, once per finished episode:  `successes.append(goal_reached)` on a `deque(maxlen=100)`, then promote when
`sum(successes) / 100 > 0.96` — also while the deque is still filling (the divisor is the limit, not the length).
N envs run episodes concurrently, so the order the deque sees has to be defined.  In the reference the order in which
episodes start and the order in which they finish are the same thing; here they are not, and the COMPLETION order is biased:
all envs of a level start together, so the first episodes to finish are the short ones — at level 0 the quick successes
(measured: a 4 096-env run "passes" 97/100 after 24 agent periods with a 37 % policy).  `EpisodeOrder` therefore feeds the
deque generation by generation: every judged env's first episode of the level (by global env index), then every env's
second episode, and so on — an order that does not depend on how long episodes take.  (Ordering by the period an episode
STARTS in was tried and is still biased: envs whose previous episode was a quick success restart together and, with the
platform's phase carrying over, tend to succeed again — levels were "passed" at a 40–82 % large-sample rate.)  The engine's
episode log (include/dql.h: per agent period and wave, who finished and who finished in the goal state) is all this needs.
A generation enters the stream once every judged env has finished it (time-outs bound the wait to t_max * f_ag periods per
episode).  For one env this is the reference's sequence exactly.  `PromotionWindow` runs the deque rule on an ordered 0/1
stream without materialising the deque: only the positions of the FAILED episodes matter.

A window of `window` consecutive completions holds at most `max_fail` failures  <=>  two failures that are `max_fail + 1`
apart in the failure list are more than `window` positions apart, where the not-yet-existing entries before the first
episode count as failures (the reference divides by 100 from the first episode on) and the current end of the stream is a
boundary the window cannot cross.
"""
from __future__ import annotations

import numpy as np

_BITS = np.arange(64, dtype=np.uint64)
_LOW = (np.uint64(1) << _BITS) - np.uint64(1)  # mask of the bits below bit b


def failure_positions(done: np.ndarray, goal: np.ndarray):
    """done, goal: uint64[..., n_waves] masks in completion order.  Returns (positions of the failed episodes in the ordered
    completion stream (ascending int64), number of completions, completions per leading row)."""
    done = np.ascontiguousarray(done, dtype=np.uint64)
    goal = np.ascontiguousarray(goal, dtype=np.uint64)
    rows = done.shape[0] if done.ndim > 1 else 1
    d = done.reshape(-1)
    f = d & ~goal.reshape(-1)
    cnt = np.bitwise_count(d).astype(np.int64)
    offs = np.cumsum(cnt) - cnt
    nz = np.flatnonzero(f)
    if nz.size:
        bits = ((f[nz, None] >> _BITS) & np.uint64(1)).astype(bool)
        rank = np.bitwise_count(d[nz, None] & _LOW).astype(np.int64)
        pos = (offs[nz, None] + rank)[bits]
    else:
        pos = np.zeros(0, dtype=np.int64)
    per_row = cnt.reshape(rows, -1).sum(axis=1)
    return pos, int(cnt.sum()), per_row


class PromotionWindow:
    """State of the reference's success deque between chunks of agent periods."""

    def __init__(self, window: int = 100, success_rate: float = 0.96):
        self.window = int(window)
        self.success_rate = float(success_rate)
        need = next((s for s in range(self.window + 1) if s / self.window > self.success_rate), None)  # same float compare as the reference
        self.max_fail = None if need is None else self.window - need
        self.reset()

    def reset(self):
        """`self._successes = deque([], maxlen=...)` after a promotion (pkg/trainer.py:226-229)"""
        k = 1 if self.max_fail is None else self.max_fail + 1
        self._carry = np.arange(-k, 0, dtype=np.int64)  # the entries "before the first episode" count as failures
        self.episodes = 0

    def get_state(self):
        return {"carry": [int(v) for v in self._carry], "episodes": int(self.episodes)}

    def set_state(self, st):
        self._carry = np.asarray(st["carry"], dtype=np.int64)
        self.episodes = int(st["episodes"])

    def push_flags(self, flags: np.ndarray):
        """Feed an ordered 0/1 stream (1 = goal state reached).  Returns None or the index (0-based since the last reset) of the
        first episode at which the reference's test passes."""
        flags = np.asarray(flags).astype(bool)
        h = self._push(np.flatnonzero(~flags).astype(np.int64), int(flags.size), np.array([flags.size], dtype=np.int64))
        return None if h is None else h[0]

    def push(self, done: np.ndarray, goal: np.ndarray):
        """Feed the log of one chunk in COMPLETION order.  Returns None, or (episode_index, row) of the first completion at which
        the reference's test passes: episode_index counts finished episodes since the last reset (0-based), row is the agent
        period within this chunk."""
        return self._push(*failure_positions(done, goal))

    def _push(self, pos, n_done, per_row):
        hit = None
        if self.max_fail is not None and n_done:
            k = self.max_fail + 1
            F = np.concatenate([self._carry, pos, np.array([n_done], dtype=np.int64)])
            ok = np.flatnonzero(F[k:] - F[:-k] > self.window)
            if ok.size:
                end = int(F[ok[0]]) + self.window  # the first window that fits ends here
                row = int(np.searchsorted(np.cumsum(per_row), end, side="right"))
                hit = (self.episodes + end, row)
            self._carry = np.concatenate([self._carry, pos])[-k:] - n_done
        self.episodes += n_done
        return hit


class EpisodeOrder:
    """Completion masks of the judged envs -> the episodes' goal flags in the order the deque sees them: every judged env's
    first episode (by env index), then every env's second episode, and so on (see the module docstring)."""

    def __init__(self, n_cols: int, valid=None):
        """n_cols bit columns (64 per mask word); `valid` marks the columns that are real envs (default: all)"""
        self.n_cols = int(n_cols)
        self.valid = np.ones(self.n_cols, dtype=bool) if valid is None else np.asarray(valid, dtype=bool)
        self.reset()

    def reset(self):
        """level switch: every env re-enters through reset and starts counting its episodes again"""
        self.count = np.zeros(self.n_cols, dtype=np.int64)  # episodes each env has finished at this level
        self._po = np.zeros(0, dtype=np.int64); self._pe = np.zeros(0, dtype=np.int64); self._pg = np.zeros(0, dtype=bool)

    def get_state(self):
        return {"count": [int(v) for v in self.count], "po": [int(v) for v in self._po], "pe": [int(v) for v in self._pe],
                "pg": [int(v) for v in self._pg]}

    def set_state(self, st):
        self.count = np.asarray(st["count"], dtype=np.int64)
        self._po = np.asarray(st["po"], dtype=np.int64); self._pe = np.asarray(st["pe"], dtype=np.int64)
        self._pg = np.asarray(st["pg"], dtype=bool)

    @staticmethod
    def _bits(m):
        m = np.ascontiguousarray(m, dtype="<u8")
        return np.unpackbits(m.view(np.uint8).reshape(m.shape[0], -1), axis=1, bitorder="little")

    def push(self, done: np.ndarray, goal: np.ndarray) -> np.ndarray:
        """done, goal: uint64[P, n_cols / 64] of the next P agent periods.  Returns the goal flags (bool) of the episodes that
        became ordered by this chunk: a generation is released once every judged env has finished that many episodes."""
        bd, bg = self._bits(done), self._bits(goal)
        r, e = np.nonzero(bd)
        if r.size:
            g = bg[r, e].astype(bool)
            o = np.lexsort((r, e))            # per env in time order
            e_s, g_s = e[o], g[o]
            first = np.r_[True, e_s[1:] != e_s[:-1]]
            run_start = np.maximum.accumulate(np.where(first, np.arange(e_s.size), 0))
            ordinal = self.count[e_s] + (np.arange(e_s.size) - run_start)   # k-th episode of its env at this level (0-based)
            np.add.at(self.count, e_s, 1)
            self._po = np.concatenate([self._po, ordinal]); self._pe = np.concatenate([self._pe, e_s]); self._pg = np.concatenate([self._pg, g_s])
        frontier = self.count[self.valid].min() if self.valid.any() else 0   # generations every judged env has completed
        ready = self._po < frontier
        ro, re_, rg = self._po[ready], self._pe[ready], self._pg[ready]
        self._po, self._pe, self._pg = self._po[~ready], self._pe[~ready], self._pg[~ready]
        o = np.lexsort((re_, ro))
        return rg[o]
