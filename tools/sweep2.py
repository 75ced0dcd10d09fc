#!/usr/bin/env python3
"""Step-time vs env count / number of timed steps (DVFS, wave quantisation) for the fused step."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
for n in [int(x) for x in sys.argv[1].split(",")]:
    for steps in [int(x) for x in sys.argv[2].split(",")]:
        e = Engine(DqlConfig(dtype=F32), n, seed=42)
        e.train_steps(5, 1.0); e.sync()
        e.timer_start(); e.train_steps(steps, 1.0); ms = e.timer_stop()
        e.kernel_timer(True); e.train_steps(10, 1.0); k, kn = e.kernel_time_ms(); e.kernel_timer(False)
        print(json.dumps({"envs": n, "steps": steps, "us_per_step": ms * 1e3 / steps, "kernel_us_evt": k * 1e3}), flush=True)
        e.close()
