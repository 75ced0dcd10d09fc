import sys, time, tempfile
from pathlib import Path
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
g.build_hip()
import bench
from dql_multirotor_landing_amd.config import F32
from dql_multirotor_landing_amd.engine import Engine
from dql_multirotor_landing_amd.trainer import Trainer
full = Engine.episode_log_read
def old_read(self): return full(self)
n = 32768
for rep in range(2):
    for tag in ("words", "full"):
        Engine.episode_log_read = full if tag == "words" else old_read
        with tempfile.TemporaryDirectory() as d:
            tr = Trainer(n_envs=n, mode="paper", save_path=Path(d) / "run", dtype=F32, sync_period=bench.CURRICULUM_SYNC, max_num_episodes=384 * n, seed=42, checkpoint_every=10**9, **bench.CURRICULUM_KW)
            t0 = time.perf_counter(); h = tr.curriculum_training(); wall = time.perf_counter() - t0
            print(tag, rep, "wall %.3f" % wall, "to_stage4 %.3f" % h[3]["wall_since_start_s"], "us/period %.2f" % (wall / sum(x["agent_periods"] for x in h) * 1e6), flush=True)
            tr._engine.close()
