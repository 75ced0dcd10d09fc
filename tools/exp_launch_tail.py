#!/usr/bin/env python3
"""What a launch costs beyond its periods at the headline batch: us per launch for P = 1, 2, 4, 8, 16 periods per launch in TRAIN mode (the writer
workgroups fold the previous launch's accumulators) and in EVAL mode (greedy, nothing to fold: the writers only publish) — intercept and slope."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
kw = dict(per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1)
for mode in ("train", "eval"):
    row = {}
    for P in (1, 2, 4, 8, 16):
        e = Engine(DqlConfig(dtype=F32, fold_per_step=1, **kw), n, seed=42)
        e.set_option("periods_per_launch", P)
        run = (lambda k: e.train_steps(k, 1.0)) if mode == "train" else (lambda k: e.eval_steps(k))
        e.train_steps(512, 1.0); run(16 * P); e.sync()
        L = 64
        e.timer_start(); run(L * P); ms = e.timer_stop()
        row[P] = ms * 1e3 / L
        e.close()
    Ps = np.array(list(row)); t = np.array([row[p] for p in row])
    slope, icpt = np.polyfit(Ps, t, 1)
    print(json.dumps({"envs": n, "mode": mode, "us_per_launch": {str(k): round(v, 2) for k, v in row.items()}, "us_per_period_slope": round(float(slope), 2), "us_per_launch_intercept": round(float(icpt), 2)}), flush=True)
