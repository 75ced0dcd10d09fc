#!/usr/bin/env python3
"""cProfile of one curriculum run with bench.py's recipe (where the Trainer's host loop spends its time between launches)."""
import cProfile, io, pstats, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g
g.build_hip()
import bench
from dql_multirotor_landing_amd.config import F32
from dql_multirotor_landing_amd.trainer import Trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
prof = len(sys.argv) > 2 and sys.argv[2] == "profile"
with tempfile.TemporaryDirectory() as d:
    tr = Trainer(n_envs=n, mode="paper", save_path=Path(d) / "run", dtype=F32, sync_period=bench.CURRICULUM_SYNC, max_num_episodes=384 * n, seed=42, checkpoint_every=10**9, **bench.CURRICULUM_KW)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if prof: pr.enable()
    h = tr.curriculum_training()
    if prof: pr.disable()
    wall = time.perf_counter() - t0
    periods = sum(x["agent_periods"] for x in h)
    print({"wall_s": wall, "agent_periods": periods, "us_per_period": wall / periods * 1e6, "to_stage4_s": h[3]["wall_since_start_s"] if len(h) > 3 else None})
    if prof:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue())
