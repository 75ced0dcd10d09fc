#!/usr/bin/env python3
"""Do the two waves of a SIMD finish a launch together?  Start / end wall clock (100 MHz) of every env wave of 16-period launches at the headline
batch (diagnostic build -DDQL_WAVE_CLOCK=7 through DQL_LIB_PATH): distribution of wave lifetimes and of the end times inside a launch.
    tools/ab_build.sh clock7 -DDQL_WAVE_CLOCK=7; DQL_LIB_PATH=.../libdql_hip_clock7.so python tools/exp_wave_tail.py [envs] [periods_per_launch]"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
P = int(sys.argv[2]) if len(sys.argv) > 2 else 16
e = Engine(DqlConfig(dtype=F32, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1, fold_per_step=1), n, seed=42)
e.set_option("periods_per_launch", P)
e.train_steps(20 * P, 1.0); e.sync()
L = 12
e.episode_log_enable(L * P)
e.timer_start(); e.train_steps(L * P, 1.0); ms = e.timer_stop()
t0, t1 = e.episode_log_read()   # the diagnostic build writes start clocks into row 0 and end clocks into row 1 of every LAUNCH's log rows
t0 = t0.astype(np.int64)[::P]; t1 = t1.astype(np.int64)[::P]
out = []
for j in range(2, L):
    s, f = (t0[j] - t0[j].min()) / 100.0, (t1[j] - t0[j].min()) / 100.0   # us since the launch's first wave started
    life = f - s
    out.append({"launch_span_us": float(f.max()), "start_p50_us": float(np.median(s)), "start_max_us": float(s.max()),
                "end_p1_us": float(np.percentile(f, 1)), "end_p25_us": float(np.percentile(f, 25)), "end_p50_us": float(np.median(f)), "end_p75_us": float(np.percentile(f, 75)),
                "end_p99_us": float(np.percentile(f, 99)), "life_p10_us": float(np.percentile(life, 10)), "life_p50_us": float(np.median(life)), "life_p90_us": float(np.percentile(life, 90))})
keys = out[0].keys()
print(json.dumps({"envs": n, "periods_per_launch": P, "us_per_launch": ms * 1e3 / L, **{k: round(float(np.mean([o[k] for o in out])), 2) for k in keys}}))
hist, edges = np.histogram((t1[5] - t0[5].min()) / 100.0, bins=24)
print(json.dumps({"end_time_histogram_us": [round(float(x), 1) for x in edges], "waves": hist.tolist()}))
e.close()
# by presumed XCD (workgroup index mod 8: the dispatcher deals workgroups round-robin over the 8 XCDs) and by presumed CU slot
e2 = None
f5 = (t1[5] - t0[5].min()) / 100.0
wg = np.arange(f5.size) // (4 if n > 8192 else 1)
print(json.dumps({"end_time_by_workgroup_mod_8_us": [round(float(f5[wg % 8 == k].mean()), 1) for k in range(8)],
                  "end_time_by_workgroup_mod_8_p99_us": [round(float(np.percentile(f5[wg % 8 == k], 99)), 1) for k in range(8)],
                  "end_time_first_vs_second_half_of_the_grid_us": [round(float(f5[: f5.size // 2].mean()), 1), round(float(f5[f5.size // 2:].mean()), 1)]}))
