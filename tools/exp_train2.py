#!/usr/bin/env python3
"""Diagnostics of curriculum training: per-chunk terminal histogram, visited-state spread, at a given level after
warm-starting level 0."""
import json, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32, Q_PAPER
from dql_multirotor_landing_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fold = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eps_hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ratios = [1.0, 0.8172650252856599, 0.8211253690681617, 0.8257273369742982, 0.8311571820651724]
eng = Engine(DqlConfig(dtype=F32, quirks=Q_PAPER, fold_per_step=fold), n, seed=42)
prev = eng.stats()
def chunk(steps, eps):
    global prev
    eng.train_steps(steps, eps)
    s = eng.stats()
    d = {k: s["by_code"][k] - prev["by_code"][k] for k in s["by_code"] if s["by_code"][k] - prev["by_code"][k]}
    ep = s["episodes"] - prev["episodes"]; dec = s["decisions"] - prev["decisions"]; rew = s["reward_sum"] - prev["reward_sum"]
    prev = s
    return ep, d, dec / max(ep, 1), rew / max(dec, 1)
for level in range(0, 3):
    if level > 0:
        eng.transfer(level, ratios[level])
    eng.set_curriculum(level)
    eng.train_steps(1, 0.0); prev = eng.stats()
    for c in range(40):
        eps = (1.0 if c < 2 else 0.01) if level == 0 else eps_hi
        ep, d, steps_per_ep, mean_r = chunk(512, eps)
        succ = d.get("TERMINAL_SUCCESS", 0) / max(ep, 1)
        if c % 4 == 3 or succ > 0.96:
            print(json.dumps({"level": level, "chunk": c, "episodes": ep, "succ": round(succ, 3), "hist": d, "steps/ep": round(steps_per_ep, 1), "mean_r": round(mean_r, 3)}), flush=True)
        if succ > 0.96:
            break
    qa, qb, cnt = eng.get_tables()
    print("level", level, "visited cells per level", [(int((cnt[k] > 0).sum()), int(cnt[k].sum())) for k in range(5)], flush=True)
