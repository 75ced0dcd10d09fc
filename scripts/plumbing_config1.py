#!/usr/bin/env python3
"""BASELINE.json configs[0]: ONE env, x-axis, curriculum step 0, 1 000 time steps through the drop-in single-env API
(`TrainingLandingEnv.reset/.step`, `DoubleQLearningAgent.guess/.update`, `Trainer.alpha/.exploration_rate`) exactly as the
reference's trainer loop drives them (pkg/trainer.py:187-212) — plumbing: Python host -> ctypes -> HIP kernels -> .npy round trip.
The reference + Gazebo needs 1/22.92 s of wall time per step (>= 43.6 s for these 1 000 steps); it cannot run here."""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

if __name__ == "__main__":
    import __graft_entry__ as g
    g.build_hip()
    from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
    from dql_multirotor_landing_amd.landing_simulation_env import TrainingLandingEnv
    from dql_multirotor_landing_amd.trainer import Trainer
    from dql_multirotor_landing_amd.config import F32, F64

    def loop(dtype, d, n_steps=1000):
        """(agent, steps, episodes, wall of the FIRST n_steps steps — library load, context creation and every kernel's first launch included —, wall of the NEXT n_steps)"""
        tr = Trainer(save_path=Path(d) / "run", n_envs=1)  # np.random.seed(42) as the reference's Trainer does
        agent = tr._double_q_learning_agent
        env = TrainingLandingEnv(0, t_max=20, f_ag=22.92, p_max=4.5, z_init=4.0, dtype=dtype)
        steps, episodes, walls = 0, 0, []
        for part in range(2):
            t0 = time.perf_counter()
            limit = (part + 1) * n_steps
            while steps < limit:
                s = env.reset(); done = False
                while not done and steps < limit:
                    a = agent.guess(s, tr.exploration_rate(episodes, 0))
                    s2, r, done, info = env.step(a)
                    sa = s + (a,)
                    agent.update(sa, s2, tr.alpha(sa), 0.99, r)
                    s = s2; steps += 1
                episodes += 1
            walls.append(time.perf_counter() - t0)
        env.close()
        return agent, steps, episodes, walls

    with tempfile.TemporaryDirectory() as d:
        agent, steps, episodes, walls = loop(F64, d)
        agent.save(Path(d))
        back = DoubleQLearningAgent.load(Path(d))
        ok = np.array_equal(back.Q_table_a, agent.Q_table_a) and np.array_equal(back.state_action_counter, agent.state_action_counter)
    with tempfile.TemporaryDirectory() as d:  # the same loop with the float32 step (TrainingLandingEnv(dtype=F32), build-specific keyword)
        _, steps32, _, walls32 = loop(F32, d)
    print(json.dumps({"config": "BASELINE configs[0]: 1 env, x-axis, level 0, 1000 steps, single-env drop-in API on the GPU",
                      "steps": 1000, "episodes": episodes, "wall_s": walls[0], "env_steps_per_s": 1000 / walls[0], "env_steps_per_s_next_1000": 1000 / walls[1],
                      "visits": float(agent.state_action_counter.sum()), "npy_round_trip_ok": bool(ok),
                      "float32_env_steps_per_s": 1000 / walls32[0], "float32_env_steps_per_s_next_1000": 1000 / walls32[1],
                      "note": "env_steps_per_s: float64 env (the default: the reference's expressions, what golden G13 pins), the process's FIRST 1000 steps: one-time work "
                              "(HIP context, the agent's resident tables, every kernel's first launch: ~15 ms) is inside; *_next_1000: the following 1000 steps of the same loop; "
                              "float32_*: TrainingLandingEnv(dtype=F32), run second in this process (the library is warm: its first 1000 carry only their own objects' creation)",
                      "reference_gazebo_env_steps_per_s": 20.18, "reference_python_mdp_plus_agent_env_steps_per_s": 14400}))
