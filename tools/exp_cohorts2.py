#!/usr/bin/env python3
"""Cohorts revisited (round 2's prototype ran at one period per launch): N envs as ONE engine on the windowed schedule vs C engines of N / C envs
driven by one host thread (dist.ShardedGroup: in-process ranks, peer-to-peer window exchange) — bit-identical results for the same sync period;
the question is whether the cohorts' launch boundaries hide behind each other's ticks.

    python tools/exp_cohorts2.py [N=131072] [cohorts=1,2,4] [P=16] [sync=16]
"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.dist import LocalWindowReducer, ShardedGroup, ShardedRunner, shard_range
from dql_multirotor_landing_amd.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
cohorts = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2", "4"])]
P = int(sys.argv[3]) if len(sys.argv) > 3 else 16
S = int(sys.argv[4]) if len(sys.argv) > 4 else 16
FAIR = int(sys.argv[5]) if len(sys.argv) > 5 else -1   # option "fair_prio" of every cohort's engine (-1: the library's automatic choice)
cfg = dict(dtype=F32, fold_per_step=1, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1)
for C in cohorts:
    engs = []
    for r in range(C):
        lo, hi = shard_range(N, r, C)
        e = Engine(DqlConfig(**cfg), hi - lo, seed=42, env_id_offset=lo)
        e.set_option("periods_per_launch", P); e.set_option("tick", 4)
        if FAIR >= 0: e.set_option("fair_prio", FAIR)
        engs.append(e)
    run = ShardedRunner(engs[0], LocalWindowReducer(engs[0]), sync_period=S) if C == 1 else ShardedGroup(engs, sync_period=S)
    run.train_steps(20 * P, 1.0); run.sync()
    for e in engs: e.sync()
    d0 = sum(e.stats()["decisions"] for e in engs)
    steps = 100 * P
    t0 = time.perf_counter()
    run.train_steps(steps, 1.0); run.sync()
    for e in engs: e.sync()
    wall = time.perf_counter() - t0
    d1 = sum(e.stats()["decisions"] for e in engs)
    print(json.dumps({"envs": N, "cohorts": C, "periods_per_launch": P, "sync_period": S, "fair_prio": FAIR, "us_per_period": wall * 1e6 / steps, "env_steps_per_s": (d1 - d0) / wall}), flush=True)
    for e in engs: e.close()
