#!/usr/bin/env python3
"""Per-chunk population success rate of every level of one curriculum run (logs/scalars.csv): is a level that does not promote flat or oscillating?
    python tools/exp_level_series.py SEED [key=value ...]"""
import csv, json, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from dql_multirotor_landing_amd.trainer import Trainer
seed = int(sys.argv[1])
kw = dict(quirks=96, judge_envs=64, periods_per_launch=16, eps_tail=0.0, eps_tail_after=192, population_gate=0.94, sync_period=16)
for a in sys.argv[2:]:
    k, v = a.split("=")
    kw[k] = None if v == "None" else (float(v) if "." in v else int(v))
envs = kw.pop("envs", 32768); budget = kw.pop("budget", 768)
with tempfile.TemporaryDirectory() as d:
    tr = Trainer(mode="paper", n_envs=envs, save_path=Path(d) / "run", chunk_steps=64, max_num_episodes=int(budget * envs), checkpoint_every=10**9, seed=seed, **kw)
    hist = tr.curriculum_training()
    rows = list(csv.DictReader(open(Path(d) / "run" / "logs" / "scalars.csv")))
    for h in hist:
        r = [x for x in rows if int(x["Curriculum step"]) == h["level"]]
        sr = [round(float(x["Episode/Success Rate"]), 3) for x in r]
        step = max(1, len(sr) // 60)
        print(json.dumps({"seed": seed, "level": h["level"], "promoted": h["promoted"], "restarts": h.get("restarts"), "chunks": len(sr), "episodes_per_env": round(h["episodes"] / envs, 1),
                          "success_rate_every_%d_chunks" % step: sr[::step]}))
    tr._engine.close()
