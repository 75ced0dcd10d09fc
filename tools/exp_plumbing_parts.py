#!/usr/bin/env python3
"""Where one iteration of the single-env loop (scripts/plumbing_config1.py) spends its time: each call of the drop-in API timed alone."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import __graft_entry__ as g
g.build_hip()
from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
from dql_multirotor_landing_amd.landing_simulation_env import TrainingLandingEnv

def t(f, n=2000):
    for _ in range(50): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6

a = DoubleQLearningAgent(5)
env = TrainingLandingEnv(0, t_max=20, f_ag=22.92, p_max=4.5, z_init=4.0)
s = env.reset()
out = {}
a.update(s + (1,), s, 0.5, 0.99, 1.0)
out["predict_answered_by_last_update_us"] = t(lambda: a.predict(s))
other = (0, 1, 1, 1, 3)
out["predict_round_trip_us"] = t(lambda: a.predict(other))
out["guess_us"] = t(lambda: a.guess(s, 0.5))
out["update_us"] = t(lambda: a.update(s + (1,), s, 0.5, 0.99, 1.0))
out["check_state_us"] = t(lambda: a._check_state(s + (1,), 6))
out["resident_us"] = t(lambda: a._resident())
def step():
    global s
    s2, r, d, info = env.step(2)
    if d: env.reset()
out["env_step_us"] = t(step, 1000)
eng = env._vec.engine
act = np.zeros(1, np.uint8)
out["engine_step_raw_plus_sync_us"] = t(lambda: (eng.step_raw(act), eng.sync()), 1000)
out["engine_step_raw_plus_outputs_us"] = t(lambda: (eng.step_raw(act), eng.step_outputs_view()), 1000)
eng.kernel_timer(True)
for _ in range(200): eng.step_raw(act); eng.sync()
out["step_kernel_us"] = eng.kernel_time_ms()[0] * 1e3
print(json.dumps(out, indent=1))
# the float64 step kernel of ONE env under each tick layout (option "tick": 0 auto, 1 plain, 2 lone, 3 packed) and for float32
from dql_multirotor_landing_amd.config import DqlConfig, F32, F64
from dql_multirotor_landing_amd.engine import Engine
for dt, name in ((F64, "f64"), (F32, "f32")):
    for tick in range(0, 5 if dt == F32 else 4):
        e = Engine(DqlConfig(dtype=dt, z_init=4.0), 1, seed=1)
        e.set_option("tick", tick)
        e.reset(None)
        for _ in range(50): e.step_raw(act); e.sync()
        e.kernel_timer(True)
        for _ in range(300): e.step_raw(act); e.sync()
        print(json.dumps({"dtype": name, "tick": tick, "step_kernel_us": e.kernel_time_ms()[0] * 1e3}))
        e.close()
