"""Several whole curricula, the first good one kept.

Why: tabular Double-Q learning from one seed is a draw.  With everything else fixed (bench.py's recipe, 32 768 envs), 48 seeds end between 0.79 and 0.94
touchdown rate although 45 of them promote all five levels by the reference's rule (profiles/r5_curriculum_48_seeds.jsonl): the rule (pkg/trainer.py:218-232)
judges the goal state of the training world, not the landing, and how well a run lands is decided below level 4 (profiles/r5_curriculum_final_candidates.jsonl) —
nothing applied to the last level moves the worst run.  A curriculum takes about two seconds here (the reference: days of Gazebo time), so the cheap cure is to
train again and keep the tables that land, judged the way the reference's `scripts/simulation.py` judges a trained agent (greedy landing episodes):
attempt j trains from seed `seed + 7919 j`; it is ACCEPTED when every level was promoted by the rule and its greedy landing score on a selection batch (a
seed of its own, not the one results are reported on) reaches `accept_touchdown`; after `max_attempts` the best attempt seen is kept and reported as not accepted.

The promotion rule stays what it was — a necessary condition of every accepted attempt."""
from __future__ import annotations

import time
from typing import Callable, Optional

import numpy as np

SEED_STRIDE = 7919
SELECTION_SEED = 977  # (bench.py and scripts/simulation.py report on seed 123)


def attempt_seed(seed: int, j: int) -> int:
    return int(seed) + SEED_STRIDE * int(j)


def _broadcast_from_rank0(comm, rank: int, v):
    """rank 0's float vector on every rank (the control plane's all-reduce: the other ranks contribute zeros)"""
    v = np.asarray(v, dtype=np.float64)
    if comm is None:
        return v
    return np.asarray(comm.all_reduce_sum(v if rank == 0 else np.zeros_like(v)), dtype=np.float64)


def curriculum_attempts(make_trainer: Callable[[int], object], score: Callable[[object], dict], max_attempts: int = 6, accept_touchdown: float = 0.875,
                        comm=None, rank: int = 0, close: Optional[Callable[[object], None]] = None, log: Optional[Callable[[dict], None]] = None) -> dict:
    """make_trainer(j) -> a fresh Trainer for attempt j (its own seed, tables and save_path; the same `comm` on every rank);
    score(trainer) -> {"touchdown_rate": .., "goal_hold_rate": ..} of the trainer's final tables, called on rank 0 only and shared with the other ranks;
    close(trainer): called once the attempt has been scored (default: close its engine).
    Returns {"chosen": j, "accepted": bool, "trainer": the chosen attempt's trainer, "history": its history, "attempts": [one record per attempt],
    "wall_s": all attempts and their scoring}."""
    if max_attempts < 1:
        raise ValueError("max_attempts must be >= 1")
    if close is None:
        def close(tr):
            eng = getattr(tr, "_engine", None)
            if eng is not None:
                eng.close()
    t_start = time.perf_counter()
    records, trainers = [], []
    chosen, accepted = None, False
    for j in range(int(max_attempts)):
        tr = make_trainer(j)
        t0 = time.perf_counter()
        hist = tr.curriculum_training()
        wall_train = time.perf_counter() - t0
        sc = score(tr) if rank == 0 else {"touchdown_rate": 0.0, "goal_hold_rate": 0.0}
        td, gh = _broadcast_from_rank0(comm, rank, [sc["touchdown_rate"], sc["goal_hold_rate"]])
        close(tr)
        promoted = sum(1 for h in hist if h["promoted"])
        by_rule = promoted == len(hist) and len(hist) > 0
        rec = {"attempt": j, "promoted_levels": promoted, "levels": len(hist), "all_levels_by_rule": by_rule,
               "selection": {"touchdown_rate": float(td), "goal_hold_rate": float(gh)}, "wall_train_s": wall_train,
               "wall_since_start_s": time.perf_counter() - t_start,
               # when THIS attempt entered the last level with every level before it promoted by the rule (None: it never did), on the clock of the whole call
               "wall_last_level_by_rule_s": None}
        if len(hist) > 1 and all(h["promoted"] for h in hist[:-1]):
            h3 = hist[-2]
            rec["wall_last_level_by_rule_s"] = (t0 - t_start) + float(h3.get("wall_first_promoted_s") or h3["wall_since_start_s"])
        records.append(rec)
        trainers.append((tr, hist))
        if log is not None:
            log(rec)
        if by_rule and td >= accept_touchdown:
            chosen, accepted = j, True
            break
    if chosen is None:  # nothing accepted: most levels by the rule first, then the landing
        chosen = max(range(len(records)), key=lambda k: (records[k]["promoted_levels"], records[k]["selection"]["touchdown_rate"], -k))
    tr, hist = trainers[chosen]
    return {"chosen": chosen, "accepted": accepted, "trainer": tr, "history": hist, "attempts": records, "wall_s": time.perf_counter() - t_start,
            "accept_touchdown": accept_touchdown, "max_attempts": int(max_attempts)}
