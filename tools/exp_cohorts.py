#!/usr/bin/env python3
"""Do the memory phases of one cohort of envs hide under the tick loops of another?  K engines (own HIP stream each) share the
GPU, each with n / K envs, launched round-robin; cohort c starts `offset` of a step late.  Aggregate env-steps/s vs one engine."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine

def run(n, k, offset_us, steps, block=0):
    engs = [Engine(DqlConfig(dtype=F32), n // k, seed=42, env_id_offset=c * (n // k)) for c in range(k)]
    for e in engs:
        e.set_option("block", block); e.train_steps(30, 1.0)
    for e in engs:
        e.sync()
    d0 = sum(e.stats()["decisions"] for e in engs)
    for c, e in enumerate(engs):
        if c and offset_us:
            e.delay(c * offset_us)
    t0 = time.perf_counter()
    for _ in range(steps):
        for e in engs:
            e.train_steps(1, 1.0)
    t_issue = time.perf_counter() - t0
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    d1 = sum(e.stats()["decisions"] for e in engs)
    for e in engs:
        e.close()
    return {"envs": n, "cohorts": k, "offset_us": offset_us, "block": block, "us_per_step_all_envs": dt * 1e6 / steps, "host_issue_us_per_step": t_issue * 1e6 / steps,
            "env_steps_per_s": (d1 - d0) / dt}

if __name__ == "__main__":
    sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "32768,65536,131072,262144".split(","))]
    for n in sizes:
        steps = max(100, min(1000, int(2e7 // n)))
        base = run(n, 1, 0, steps)
        print(json.dumps(base), flush=True)
        for k in (2, 4):
            for off in (0, base["us_per_step_all_envs"] / k):
                print(json.dumps(run(n, k, off, steps)), flush=True)
