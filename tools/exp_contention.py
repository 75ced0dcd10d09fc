#!/usr/bin/env python3
"""Step time when the envs crowd into few table cells (a trained greedy policy holding the goal state) vs spread out (eps = 1):
the accumulation path of the step kernel must not degrade under same-address atomics.

    [DQL_LIB_PATH=...] python tools/exp_contention.py
"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from dql_multirotor_landing_amd.config import Q_PAPER, training_config, F32
from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
from dql_multirotor_landing_amd.engine import Engine
agent = DoubleQLearningAgent.load(ROOT / "tests" / "golden" / "assets")
for n in (4096, 65536):
    for block in (64, 256):
        for name, eps, level in (("spread eps=1 level 0", 1.0, 0), ("greedy reference tables level 4", 0.0, 4)):
            e = Engine(training_config(level, dtype=F32, quirks=Q_PAPER), n, seed=42)
            e.set_option("block", block)
            e.set_tables(*agent._padded())
            e.train_steps(300, eps); e.sync()
            idx = e.states()
            s0 = e.stats(); e.timer_start(); e.train_steps(500, eps); ms = e.timer_stop(); s1 = e.stats()
            cells = np.unique(idx, return_counts=True)
            print(json.dumps({"envs": n, "block": block, "regime": name, "us_per_step": ms * 1e3 / 500, "distinct_states": int(cells[0].size),
                              "largest_state_share": float(cells[1].max() / n)}), flush=True)
            e.close()
