#!/usr/bin/env python3
"""Why do greedy episodes time out at the upper curriculum levels?  Per-step discrete states of 2048 greedy envs at `level`
(training flavour, paper quirks): longest in-goal streak per episode, what breaks a streak (position bin, velocity bin, level)."""
import json, sys
from collections import Counter
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from dql_multirotor_landing_amd.config import Q_PAPER, training_config
from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
from dql_multirotor_landing_amd.engine import Engine
tables = sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "tests" / "golden" / "assets")
level = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 2048
agent = DoubleQLearningAgent.load(Path(tables))
eng = Engine(training_config(level, quirks=Q_PAPER), n, seed=7)
eng.set_tables(*agent._padded())
eng.eval_steps(1)
names = eng.field_names(); inames = eng.field_names(True)
T = 470
idx = np.zeros((T, n), dtype=np.int32); done_at = np.full(n, -1); code = np.full(n, -1)
relp = np.zeros((T, n), dtype=np.float32); relv = np.zeros((T, n), dtype=np.float32); act = np.zeros((T, n), dtype=np.int8)
for t in range(T):
    eng.eval_steps(1)
    reals, ints = eng.get_fields()
    idx[t] = ints[inames.index("idx_x")]; relp[t] = reals[names.index("obs_p_x")]; relv[t] = reals[names.index("obs_v_x")]; act[t] = ints[inames.index("action")] & 3
    d = (ints[inames.index("flags")] & 1) != 0
    new = d & (done_at < 0)
    done_at[new] = t; code[new] = ints[inames.index("code")][new]
k = idx // 189; pb = (idx // 63) % 3; vb = (idx // 21) % 3; ab = (idx // 7) % 3; th = idx % 7
goal = (k == level) & (pb == 1) & (vb == 1)
res = {"tables": tables, "level": level, "outcomes": dict(Counter(code.tolist()))}
to = np.flatnonzero(code == 6)[:400]
streaks, frac_goal, breakers = [], [], Counter()
for e in to:
    g = goal[:done_at[e] + 1, e]
    best = cur = 0
    for t, v in enumerate(g):
        if v:
            cur += 1; best = max(best, cur)
        else:
            if cur > 0:  # what ended the streak
                breakers["level" if k[t, e] != level else ("pos" if pb[t, e] != 1 else "vel")] += 1
            cur = 0
    streaks.append(best); frac_goal.append(g.mean())
res["timeout_episodes"] = len(to)
if len(to):
    res["longest_goal_streak"] = {"mean": float(np.mean(streaks)), "p50": float(np.median(streaks)), "p90": float(np.percentile(streaks, 90)), "max": int(np.max(streaks))}
    res["fraction_of_steps_in_goal"] = float(np.mean(frac_goal))
    res["streak_breakers"] = dict(breakers)
    e = to[0]
    res["example"] = {"p": [round(float(x), 2) for x in relp[200:260, e]], "v": [round(float(x), 2) for x in relv[200:260, e]],
                      "goal": [int(x) for x in goal[200:260, e]], "a": [int(x) for x in act[200:260, e]], "theta_bin": [int(x) for x in th[200:260, e]],
                      "acc_bin": [int(x) for x in ab[200:260, e]]}
print(json.dumps(res))
