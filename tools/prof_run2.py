#!/usr/bin/env python3
"""Like prof_run.py with a configurable agent rate (ticks per period) to separate per-period from per-tick instruction counts:
prof_run2.py N S F_AG [P] [cfg4]   (periods per launch P, default 1)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
n = int(sys.argv[1]); steps = int(sys.argv[2]); f_ag = float(sys.argv[3])
ppl = int(sys.argv[4]) if len(sys.argv) > 4 else 1
kw = dict(per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1) if len(sys.argv) > 5 and sys.argv[5] == "cfg4" else {}
e = Engine(DqlConfig(dtype=F32, f_ag=f_ag, t_max=10000.0 / f_ag, fold_per_step=1, **kw), n, seed=42)
e.set_option("periods_per_launch", ppl)
e.train_steps(steps, 1.0); e.sync()
print(e.stats())
