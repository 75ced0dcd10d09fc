#!/usr/bin/env python3
"""Fixed cost of one launch vs cost per physics tick: agent rate varied so that a period has 1, 5, 11, 22 ticks."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # periods per launch
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4096, 1048576]
for n in sizes:
    for f_ag in (500.0, 100.0, 45.4545, 22.92):
        e = Engine(DqlConfig(dtype=F32, f_ag=f_ag, t_max=100.0), n, seed=1)
        e.set_option("periods_per_launch", P)
        e.train_steps(20, 1.0); e.sync()
        steps = max(48, min(400, int(2e7 // n))) // 8 * 8
        e.timer_start(); e.train_steps(steps, 1.0); ms = e.timer_stop()
        t = e.stats()["physics_ticks"] / e.stats()["agent_steps"]
        print(json.dumps({"envs": n, "periods_per_launch": P, "ticks_per_period": round(t, 2), "us_per_step": ms * 1e3 / steps}), flush=True)
        e.close()
