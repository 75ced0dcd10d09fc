#!/usr/bin/env python3
"""Does a longer episode budget per level change the stage-4 policy?  bench.py's recipe, budget = B episodes per env and level.
    python tools/exp_budget.py 32768 6 384,768,1536"""
import json, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "scripts"))
import __graft_entry__ as g
g.build_hip()
import bench, simulation
from dql_multirotor_landing_amd.config import F32, Q_PAPER
from dql_multirotor_landing_amd.trainer import Trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
budgets = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "384,768").split(",")]
for B in budgets:
    for seed in bench.CURRICULUM_SEEDS[:n_seeds]:
        with tempfile.TemporaryDirectory() as d:
            tr = Trainer(mode="paper", n_envs=n, dtype=F32, save_path=Path(d) / "run", chunk_steps=64, sync_period=bench.CURRICULUM_SYNC, max_num_episodes=B * n,
                         checkpoint_every=10**9, seed=seed, **bench.CURRICULUM_KW)
            t0 = time.perf_counter(); h = tr.curriculum_training(); wall = time.perf_counter() - t0
            ev = simulation.evaluate(Path(d) / "run", 4096, 4, flavour="training", quirks=Q_PAPER)
            td = simulation.evaluate(Path(d) / "run", 4096, 4, flavour="simulation", quirks=Q_PAPER)
            tr._engine.close()
        print(json.dumps({"budget_episodes_per_env_and_level": B, "seed": seed, "envs": n, "promoted": [bool(x["promoted"]) for x in h],
                          "online_success_at_handover": [round(x["success_rate"], 3) for x in h], "wall_to_stage4_s": round(h[3]["wall_since_start_s"], 2), "wall_s": round(wall, 2),
                          "goal_hold_rate": ev["TERMINAL_SUCCESS"] / 4096, "touchdown_rate": td["TERMINAL_CONTACT"] / 4096}), flush=True)
