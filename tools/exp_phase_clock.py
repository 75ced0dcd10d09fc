#!/usr/bin/env python3
"""Where a launch's shader cycles go, per wave and phase (diagnostic build -DDQL_PHASE_CLOCK, loaded through DQL_LIB_PATH):

    tools/ab_build.sh phase -DDQL_PHASE_CLOCK
    DQL_LIB_PATH=$PWD/dql_multirotor_landing_amd/csrc/libdql_hip_phase.so python tools/exp_phase_clock.py 4096,131072,1048576 [cfg4]
"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import lib_source_sha16
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
NAMES = ["state load", "period begin", "physics ticks", "manager ticks", "period end", "accumulate", "store + flush"]
P = 16
flav = dict(per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1) if len(sys.argv) > 2 and sys.argv[2] == "cfg4" else {}
for n in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "131072"])]:
    e = Engine(DqlConfig(dtype=F32, fold_per_step=1, **flav), n, seed=42)
    e.set_option("periods_per_launch", P)
    e.train_steps(20 * P, 1.0); e.sync()
    e.episode_log_enable(P)
    tot = np.zeros(7); reps = 6
    ms = 0.0
    for _ in range(reps):
        e.timer_start(); e.train_steps(P, 1.0); ms += e.timer_stop()
        d, g = e.episode_log_read()   # phase k of wave w sits in (done if k even else goal)[k // 2][w]
        ph = np.stack([(d if k % 2 == 0 else g)[k // 2] for k in range(7)]).astype(np.float64)
        tot += ph.mean(axis=1)
    tot /= reps
    per = tot / P
    print(json.dumps({"envs": n, "flavour": "cfg4" if flav else "shared platform", "periods_per_launch": P, "launch_us": ms * 1e3 / reps,
                      "cycles_per_wave_per_period": {NAMES[k]: round(per[k], 1) for k in range(7)}, "share": {NAMES[k]: round(tot[k] / tot.sum(), 4) for k in range(7)},
                      "cycles_per_wave_per_launch": round(tot.sum(), 0), "implied_clock_ghz": round(tot.sum() / (ms * 1e6 / reps), 3), "source_sha16": lib_source_sha16()}), flush=True)
    e.close()
