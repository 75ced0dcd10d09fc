#!/usr/bin/env python3
"""Throughput sweep of the fused step on one GPU: envs x block x lds_tables x dtype (runs on the GPU box)."""
import itertools, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32, F64
from dql_multirotor_landing_amd.engine import Engine

def run(n, block, lds, dtype, steps):
    e = Engine(DqlConfig(dtype=dtype), n, seed=42)
    e.set_option("block", block)
    e.train_steps(30, 1.0); e.sync()
    s0 = e.stats(); e.timer_start()
    e.train_steps(steps, 1.0)
    ms = e.timer_stop(); s1 = e.stats()
    d = s1["decisions"] - s0["decisions"]
    e.close()
    return {"envs": n, "block": block, "lds": lds, "dtype": "f32" if dtype == F32 else "f64", "us_per_step": ms * 1e3 / steps,
            "env_steps_per_s": d / (ms * 1e-3)}

if __name__ == "__main__":
    envs = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "4096,65536,262144,1048576".split(","))]
    blocks = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "64,256".split(","))]
    ldss = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,1".split(","))]
    dts = [F32 if x == "f32" else F64 for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["f32"])]
    for n, b, l, d in itertools.product(envs, blocks, ldss, dts):
        steps = max(20, min(1000, int(4e7 // n)))
        print(json.dumps(run(n, b, l, d, steps)), flush=True)
