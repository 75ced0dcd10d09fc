#!/usr/bin/env python3
"""Like prof_run.py with a configurable agent rate (ticks per period) to separate per-launch from per-tick instruction counts."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
n = int(sys.argv[1]); steps = int(sys.argv[2]); f_ag = float(sys.argv[3])
e = Engine(DqlConfig(dtype=F32, f_ag=f_ag, t_max=100.0), n, seed=42)
e.train_steps(steps, 1.0); e.sync()
print(e.stats())
