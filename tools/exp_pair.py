#!/usr/bin/env python3
"""Time per agent period at the bench's flavour for a list of (block, fair_prio) settings on one context size (library through DQL_LIB_PATH).
    python tools/exp_pair.py [envs] [block:fair,block:fair,...]"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
sets = [tuple(int(v) for v in x.split(":")) for x in (sys.argv[2] if len(sys.argv) > 2 else "0:-1,512:1,512:0").split(",")]
for rnd in range(2):
    for block, fair in sets:
        e = Engine(DqlConfig(dtype=F32, fold_per_step=1, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1), n, seed=42)
        e.set_option("periods_per_launch", 16); e.set_option("block", block); e.set_option("fair_prio", fair)
        e.train_steps(512, 1.0); e.sync()
        ts = []
        for _ in range(3):
            e.sync(); t0 = time.perf_counter(); e.train_steps(2000, 1.0); e.sync(); ts.append((time.perf_counter() - t0) * 1e6 / 2000)
        print(json.dumps({"envs": n, "block": block, "fair_prio": fair, "us_per_period_median": round(float(np.median(ts)), 3), "min": round(min(ts), 3), "max": round(max(ts), 3)}), flush=True)
        e.close()
