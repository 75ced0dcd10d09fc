#!/usr/bin/env python3
"""Where the step time goes at small batches: wave start / end times (100 MHz wall clock) of every env wave inside each launch
(diagnostic build -DDQL_WAVE_CLOCK, loaded through DQL_LIB_PATH), next to the per-step time of the same run.

    for k in 2 3 4 5 6 7; do tools/ab_build.sh clock$k -DDQL_WAVE_CLOCK=$k; done
    DQL_LIB_PATH=$PWD/dql_multirotor_landing_amd/csrc/libdql_hip_clock7.so python tools/exp_wave_clock.py
Phase k ends when: 2 state loaded, 3 action chosen + set-point matrix (Q-table reads), 4 tick loop done, 5 MDP + TD target done,
6 state stored + LDS accumulation issued, 7 wave end (all memory operations complete); inside phase 5: 41 new state index known (rotation,
Euler pitch, discretise; table row requested), 42 check + reward done (lanes that did not reset).
"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
for n in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "65536"])]:
    for f_ag in (22.92, 500.0):
        e = Engine(DqlConfig(dtype=F32, f_ag=f_ag, t_max=100.0), n, seed=42)
        e.train_steps(100, 1.0); e.sync()
        K = 200
        e.episode_log_enable(K)
        e.timer_start(); e.train_steps(K, 1.0); ms = e.timer_stop()
        t0, t1 = e.episode_log_read()  # "done" rows = start clocks, "goal" rows = end clocks (10 ns units)
        t0 = t0.astype(np.int64); t1 = t1.astype(np.int64)
        life = (t1 - t0) * 10.0  # ns
        first, last = t0.min(axis=1), t1.max(axis=1)
        span = (last - first) * 10.0            # first wave start -> last wave end inside a launch
        gap = (first[1:] - last[:-1]) * 10.0    # last wave end of launch j -> first wave start of launch j+1
        period = (first[1:] - first[:-1]) * 10.0
        print(json.dumps({"envs": n, "ticks_per_period": round(500.0 / f_ag, 2), "us_per_step": ms * 1e3 / K,
                          "wave_lifetime_us": {"mean": life.mean() / 1e3, "p10": float(np.percentile(life, 10)) / 1e3, "p90": float(np.percentile(life, 90)) / 1e3},
                          "launch_span_us": span.mean() / 1e3, "gap_between_launches_us": gap.mean() / 1e3, "period_us": period.mean() / 1e3,
                          "start_skew_us": float((t0.max(axis=1) - first).mean()) * 10 / 1e3}), flush=True)
        e.close()
