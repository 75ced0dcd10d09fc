"""Greedy evaluation of a set of tables in the landing flavour (`SimulationLandingEnv`'s world, pkg/landing_simulation_env.py:285-428: descent at
-0.4 m/s from z = 4 m, uniform start, no goal branch — an episode ends on the platform, outside the fly zone or on the ground), as ONE batched run
on the engine: `periods` greedy agent periods of `n_envs` envs, outcome counts of every episode that ended in them.  `scripts/simulation.py` is the
evaluation harness proper (first episode of every env); this is the cheaper score the Trainer's `final_candidates` selection uses."""
from __future__ import annotations

import numpy as np

from .config import F32, simulation_config


def landing_score(engine_cls, tables, level: int = 4, n_envs: int = 4096, periods: int = 400, seed: int = 777, dtype: int = F32, device=None, **cfg_kw):
    """{"touchdown_rate", "episodes", "by_code"} of greedy flights with `tables` = (Q_table_a, Q_table_b, state_action_counter)"""
    cfg = simulation_config(working_curriculum_step=level, dtype=dtype, **cfg_kw)
    eng = engine_cls(cfg, n_envs, seed=seed) if device is None else engine_cls(cfg, n_envs, seed=seed, device=device)
    try:
        eng.set_tables(*(np.asarray(t, dtype=np.float64).reshape(-1) for t in tables))
        eng.eval_steps(int(periods))
        s = eng.stats()
    finally:
        close = getattr(eng, "close", None)
        if close is not None:
            close()
    by = s["by_code"]
    if not isinstance(by, dict):  # (an engine that reports the histogram as a list in CheckResult order)
        from .config import CHECK_NAMES
        by = {CHECK_NAMES[i]: v for i, v in enumerate(by)}
    n = int(s["episodes"])
    return {"touchdown_rate": (by.get("TERMINAL_CONTACT", 0) / n) if n else 0.0, "episodes": n, "by_code": {k: int(v) for k, v in by.items() if v}}
