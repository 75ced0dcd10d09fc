#!/usr/bin/env python3
"""scripts/plumbing_config1.py's loop with a clock around every call of the drop-in API: where the per-step time of the single-env loop goes IN the loop
(tools/exp_plumbing_parts.py times each call alone)."""
import json, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g
g.build_hip()
from dql_multirotor_landing_amd.landing_simulation_env import TrainingLandingEnv
from dql_multirotor_landing_amd.trainer import Trainer
from dql_multirotor_landing_amd.config import F32, F64
pc = time.perf_counter
for dtype, name in ((F64, "f64"), (F32, "f32")):
    with tempfile.TemporaryDirectory() as d:
        tr = Trainer(save_path=Path(d) / "run", n_envs=1)
        agent = tr._double_q_learning_agent
        env = TrainingLandingEnv(0, t_max=20, f_ag=22.92, p_max=4.5, z_init=4.0, dtype=dtype)
        acc = dict(reset=0.0, eps=0.0, rng=0.0, guess=0.0, step=0.0, tuple=0.0, alpha=0.0, update=0.0)
        steps = episodes = 0
        t_all = pc()
        while steps < 1000:
            t0 = pc(); s = env.reset(); acc["reset"] += pc() - t0; done = False
            while not done and steps < 1000:
                t0 = pc(); e = tr.exploration_rate(episodes, 0); t1 = pc(); explore = np.random.uniform(0, 1) < e; rnd = np.random.randint(3); t1b = pc(); greedy = agent.predict(s); a = int(rnd) if explore else greedy; t2 = pc()
                s2, r, done, info = env.step(a); t3 = pc()
                sa = s + (a,); t4 = pc(); al = tr.alpha(sa); t5 = pc()
                agent.update(sa, s2, al, 0.99, r); t6 = pc()
                acc["eps"] += t1 - t0; acc["rng"] += t1b - t1; acc["guess"] += t2 - t1b; acc["step"] += t3 - t2; acc["tuple"] += t4 - t3; acc["alpha"] += t5 - t4; acc["update"] += t6 - t5
                s = s2; steps += 1
            episodes += 1
        wall = pc() - t_all
        env.close()
        print(json.dumps({"dtype": name, "us_per_step": round(wall * 1e3, 2), "episodes": episodes, **{k + "_us": round(v * 1e3, 2) for k, v in acc.items()},
                          "unaccounted_us": round((wall - sum(acc.values())) * 1e3, 2)}))
