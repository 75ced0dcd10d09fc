import json, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
for n in (4096, 32768, 65536, 98304):
    for rnd in range(2):
        for tick, block in ((0, 0), (4, 0), (4, 256), (4, 64)):
            e = Engine(DqlConfig(dtype=F32, fold_per_step=1, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1), n, seed=42)
            e.set_option("periods_per_launch", 16); e.set_option("tick", tick); e.set_option("block", block)
            e.train_steps(512, 1.0); e.sync()
            ts = []
            for _ in range(3):
                e.sync(); t0 = time.perf_counter(); e.train_steps(2000, 1.0); e.sync(); ts.append((time.perf_counter() - t0) * 1e6 / 2000)
            print(json.dumps({"envs": n, "tick": tick, "block": block, "us_per_period": round(float(np.median(ts)), 3)}), flush=True)
            e.close()
