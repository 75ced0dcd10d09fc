#!/usr/bin/env python3
"""What the launch's constant is made of: per-phase shader cycles of the env waves for 1 / 2 / 4 / 16 periods per launch at the headline batch, split by the
wave's role on its SIMD (first = workgroup in the first half of the grid: the older wave of its SIMD; second = the younger one).  Diagnostic build
-DDQL_PHASE_CLOCK through DQL_LIB_PATH:

    tools/ab_build.sh phase -DDQL_PHASE_CLOCK
    DQL_LIB_PATH=$PWD/dql_multirotor_landing_amd/csrc/libdql_hip_phase.so python tools/exp_phase_by_role.py [envs] [fair_prio -1/0/1]
"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
NAMES = ["state load", "period begin", "physics ticks", "manager ticks", "period end", "accumulate", "store + flush"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
fair = int(sys.argv[2]) if len(sys.argv) > 2 else -1
rows = []
for P in (4, 8, 16):   # (the log holds a launch's 7 rows of phase totals from 4 periods up)
    e = Engine(DqlConfig(dtype=F32, fold_per_step=1, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1), n, seed=42)
    e.set_option("periods_per_launch", P)
    e.set_option("fair_prio", fair)
    e.train_steps(320, 1.0); e.sync()
    e.episode_log_enable(P)
    reps = 8
    tot = np.zeros((2, 7)); ms = 0.0
    for _ in range(reps):
        e.timer_start(); e.train_steps(P, 1.0); ms += e.timer_stop()
        d, g = e.episode_log_read()
        ph = np.stack([(d if k % 2 == 0 else g)[k // 2] for k in range(7)]).astype(np.float64)   # [7, waves]
        h = ph.shape[1] // 2
        tot[0] += ph[:, :h].mean(axis=1); tot[1] += ph[:, h:].mean(axis=1)
    tot /= reps
    rows.append((P, ms * 1e3 / reps, tot.copy()))
    print(json.dumps({"envs": n, "fair_prio": fair, "periods_per_launch": P, "launch_us": round(ms * 1e3 / reps, 2),
                      **{role: {"cycles_per_launch": round(float(tot[r].sum())), **{NAMES[k]: round(float(tot[r, k])) for k in range(7)}} for r, role in enumerate(("first", "second"))}}), flush=True)
    e.close()
# per phase and role: cycles = constant + slope x periods (least squares over the three launch lengths)
Ps = np.array([r[0] for r in rows], dtype=np.float64)
A = np.stack([np.ones_like(Ps), Ps], axis=1)
for r, role in enumerate(("first", "second")):
    fit = {NAMES[k]: [round(float(x)) for x in np.linalg.lstsq(A, np.array([row[2][r, k] for row in rows]), rcond=None)[0]] for k in range(7)}
    tot_fit = [round(float(x)) for x in np.linalg.lstsq(A, np.array([row[2][r].sum() for row in rows]), rcond=None)[0]]
    print(json.dumps({"role": role, "cycles_constant_and_per_period": fit, "wave_total": tot_fit}))
us = np.linalg.lstsq(A, np.array([row[1] for row in rows]), rcond=None)[0]
print(json.dumps({"launch_us_constant": round(float(us[0]), 2), "launch_us_per_period": round(float(us[1]), 2)}))
