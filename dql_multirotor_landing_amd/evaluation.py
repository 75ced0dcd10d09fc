"""Greedy evaluation of a set of tables on the engine: the counterpart of the reference's `scripts/simulation.py` loop (load tables, fly landing
episodes, count how they end — SURVEY.md section 8 f2) for a whole batch of envs at once.

`first_episode_outcomes` is the harness (`scripts/simulation.py` and bench.py report with it); `landing_score` is the two figures the repo quotes for a
set of tables: the touchdown rate in `SimulationLandingEnv`'s world (pkg/landing_simulation_env.py:285-428: descent at -0.4 m/s from z = 4 m, an episode
ends on the platform, outside the fly zone or on the ground) and the goal-hold rate in `TrainingLandingEnv`'s (:167-283, the world the promotion rule judges)."""
from __future__ import annotations

import numpy as np

from .config import CHECK_NAMES, F32, Q_PAPER, simulation_config, training_config
from .engine import Engine


def first_episode_outcomes(tables, n_envs: int = 4096, level: int = 4, max_steps: int = 600, seed: int = 123, dtype=None, flavour: str = "simulation",
                           device=0, **cfg_kw):
    """Greedy roll-outs of `tables` = (Q_table_a, Q_table_b, state_action_counter), flat and padded as `DoubleQLearningAgent._padded()` returns them;
    the terminal histogram of the FIRST episode of every env (+ "unfinished")."""
    dtype = F32 if dtype is None else dtype
    if flavour == "simulation":
        cfg = simulation_config(working_curriculum_step=level, dtype=dtype, **cfg_kw)
    elif flavour == "training":
        cfg = training_config(level, dtype=dtype, **cfg_kw)
    else:
        raise ValueError("flavour must be 'simulation' or 'training'")
    eng = Engine(cfg, n_envs, seed=seed, device=device)
    try:
        eng.set_tables(*tables)
        first_code = np.full(n_envs, -1, dtype=np.int64)
        eng.eval_steps(1)  # reset period
        for _ in range(max_steps):
            eng.eval_steps(1)
            d, c = eng.dones()
            new = (d != 0) & (first_code < 0)
            first_code[new] = c[new]
            if (first_code >= 0).all():
                break
    finally:
        eng.close()
    hist = {CHECK_NAMES[k]: int((first_code == k).sum()) for k in range(len(CHECK_NAMES))}
    hist["unfinished"] = int((first_code < 0).sum())
    return hist


def landing_score(tables, n_envs: int = 4096, level: int = 4, seed: int = 123, dtype=None, device=0, quirks: int = Q_PAPER):
    """{"touchdown_rate", "goal_hold_rate"} of `n_envs` greedy first episodes each (the figures of bench.py's `stage4_greedy_4096_episodes`)"""
    h = first_episode_outcomes(tables, n_envs, level, seed=seed, dtype=dtype, flavour="simulation", device=device, quirks=quirks)
    g = first_episode_outcomes(tables, n_envs, level, seed=seed, dtype=dtype, flavour="training", device=device, quirks=quirks)
    return {"touchdown_rate": h["TERMINAL_CONTACT"] / n_envs, "goal_hold_rate": g["TERMINAL_SUCCESS"] / n_envs}
