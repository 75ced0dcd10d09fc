import json, sys
sys.path.insert(0, "/root/repo")
from dql_multirotor_landing_amd.config import DqlConfig, F64
from dql_multirotor_landing_amd.engine import Engine
for n in (1, 4096, 131072):
    e = Engine(DqlConfig(dtype=F64), n, seed=1)
    e.set_option("periods_per_launch", 1 if n == 1 else 16)
    e.train_steps(32, 1.0); e.sync()
    steps = 320 if n > 1 else 400
    e.timer_start(); e.train_steps(steps, 1.0); ms = e.timer_stop()
    print(json.dumps({"envs": n, "f64_us_per_period": ms * 1e3 / steps})); e.close()
