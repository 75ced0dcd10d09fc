#!/usr/bin/env python3
"""Step time per agent period vs env count and periods per launch (option "periods_per_launch"), one engine, one stream."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
envs = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "4096,16384,32768,65536,131072,262144,1048576".split(","))]
ppls = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,4".split(","))]
ticks = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0".split(","))]   # option "tick": 0 auto, 1 plain, 2 VGPR constants, 3 packed
blocks = [int(x) for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else "0".split(","))]
import itertools
for n in envs:
    for P, tick, block in itertools.product(ppls, ticks, blocks):
        e = Engine(DqlConfig(dtype=F32), n, seed=42)
        e.set_option("periods_per_launch", P); e.set_option("tick", tick); e.set_option("block", block)
        steps = max(24, min(1000, int(4e7 // n))) // 4 * 4
        e.train_steps(32, 1.0); e.sync()
        s0 = e.stats(); e.timer_start(); e.train_steps(steps, 1.0); ms = e.timer_stop(); s1 = e.stats()
        print(json.dumps({"envs": n, "periods_per_launch": P, "tick": tick, "block": block, "us_per_period": ms * 1e3 / steps, "env_steps_per_s": (s1["decisions"] - s0["decisions"]) / (ms * 1e-3)}), flush=True)
        e.close()
