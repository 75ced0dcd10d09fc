#!/usr/bin/env python3
"""Counterpart of the reference's scripts/simulation.py (load tables, run greedy landing episodes): evaluates a
pair of Q tables in the vectorised simulator and reports how the episodes end.

    python scripts/simulation.py [--tables DIR] [--envs 4096] [--level 4] [--flavour simulation|training] [--mode paper|reference]
Default tables: tests/golden/assets (a data copy of the reference's stage-4 policy).
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def evaluate(tables_dir, n_envs=4096, level=4, max_steps=600, seed=123, dtype=None, flavour="simulation", device=0, **cfg_kw):
    """Greedy roll-outs of the tables saved in `tables_dir`; returns the terminal histogram of the FIRST episode of every env."""
    from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
    from dql_multirotor_landing_amd.evaluation import first_episode_outcomes
    agent = DoubleQLearningAgent.load(Path(tables_dir))
    return first_episode_outcomes(agent._padded(), n_envs, level, max_steps, seed, dtype, flavour, device, **cfg_kw)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", default=str(Path(__file__).resolve().parent.parent / "tests" / "golden" / "assets"))
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--level", type=int, default=4)
    ap.add_argument("--flavour", default="simulation", choices=["simulation", "training"])
    ap.add_argument("--mode", default="paper", choices=["paper", "reference"],
                    help="observation / MDP quirk set of the roll-outs: 'paper' (default; what scripts/training.py --mode paper trains under) or the "
                         "reference's code as it is (frozen acceleration reference B19, sticky checks B8, ...: DESIGN.md section 3)")
    a = ap.parse_args()
    import __graft_entry__ as g
    g.build_hip()
    from dql_multirotor_landing_amd.config import Q_PAPER, Q_REFERENCE
    h = evaluate(a.tables, a.envs, a.level, flavour=a.flavour, quirks=Q_PAPER if a.mode == "paper" else Q_REFERENCE)
    n = a.envs
    print(json.dumps({"tables": a.tables, "envs": n, "level": a.level, "flavour": a.flavour, "mode": a.mode, "first_episode_outcomes": h,
                      "touchdown_rate": h["TERMINAL_CONTACT"] / n, "goal_rate": h["TERMINAL_SUCCESS"] / n}, indent=1))
