#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 counter passes: N envs, S train steps [block] [periods_per_launch] [flavour].
flavour: "" (x-axis, shared platform), "cfg4" (BASELINE configs[4]: per-env platform + observation noise), "2axis"."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ppl = int(sys.argv[4]) if len(sys.argv) > 4 else 1
flavour = sys.argv[5] if len(sys.argv) > 5 else ""
kw = {}
if flavour == "cfg4":
    kw = dict(per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1)
elif flavour == "2axis":
    kw = dict(two_axis=1)
e = Engine(DqlConfig(dtype=F32, fold_per_step=1, **kw), n, seed=42)
e.set_option("block", block)
e.set_option("periods_per_launch", ppl)
e.train_steps(steps, 1.0)
e.sync()
print(e.stats())
