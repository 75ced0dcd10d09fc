#!/usr/bin/env python3
"""Wall-clock to curriculum stage 4 under the reference's own rule (ordered 100-episode deque > 0.96, or the level's
50 000-episode budget runs out: pkg/trainer.py:187,218-232), then greedy roll-outs of the resulting stage-4 tables next to
the reference's own stage-4 tables in the same simulator.

    python tools/exp_stage4.py [--envs 4096 ...] [--modes paper reference] [--budget 50000]
"""
import argparse, json, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "scripts"))
import simulation
from dql_multirotor_landing_amd.config import Q_PAPER, Q_REFERENCE
from dql_multirotor_landing_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, nargs="+", default=[4096])
ap.add_argument("--modes", nargs="+", default=["paper"])
ap.add_argument("--budget", type=int, nargs="+", default=[50000])
ap.add_argument("--chunk", type=int, default=64)
ap.add_argument("--rule", default="ordered")
ap.add_argument("--eps-floor", type=float, default=0.0)
ap.add_argument("--seeds", type=int, nargs="+", default=[42])
ap.add_argument("--fold-per-step", type=int, default=1)
ap.add_argument("--eps-scale", type=float, default=1.0)
ap.add_argument("--quirks", type=lambda x: int(x, 0), default=None, help="override the mode's quirk bits (e.g. 0x37 = reference minus the frozen acceleration reference)")
a = ap.parse_args()
for mode in a.modes:
    quirks = (Q_PAPER if mode == "paper" else Q_REFERENCE) if a.quirks is None else a.quirks
    for n in a.envs:
        for budget, seed in [(b, sd) for b in a.budget for sd in a.seeds]:
            with tempfile.TemporaryDirectory() as d:
                t0 = time.perf_counter()
                tr = Trainer(mode=mode, n_envs=n, save_path=Path(d) / "run", chunk_steps=a.chunk, max_num_episodes=budget, checkpoint_every=10**9,
                             promotion_rule=a.rule, eps_floor=a.eps_floor, seed=seed, fold_per_step=a.fold_per_step, eps_episode_scale=a.eps_scale, quirks=a.quirks)
                hist = tr.curriculum_training()
                wall = time.perf_counter() - t0
                ev = {}
                for flavour in ("training", "simulation"):
                    h = simulation.evaluate(Path(d) / "run", 4096, 4, flavour=flavour, quirks=quirks)
                    ev[flavour] = {"goal": h["TERMINAL_SUCCESS"] / 4096, "touchdown": h["TERMINAL_CONTACT"] / 4096, "flyzone_x": h["TERMINAL_FLYZONE_X"] / 4096,
                                   "timeout": h["TERMINAL_TIMEOUT"] / 4096, "min_alt": h["TERMINAL_MINIMUM_ALTITUDE"] / 4096}
            print(json.dumps({"mode": mode, "envs": n, "budget": budget, "seed": seed, "fold_per_step": a.fold_per_step, "eps_scale": a.eps_scale, "quirks": quirks, "rule": a.rule, "eps_floor": a.eps_floor, "wall_total_s": round(wall, 3),
                              "wall_to_stage4_s": round(hist[3]["wall_since_start_s"], 3) if len(hist) > 3 else None,
                              "levels": [{"level": h["level"], "promoted": h["promoted"], "exhausted": h["exhausted"], "promoted_at": h["promoted_at"],
                                          "episodes": h["episodes"], "agent_periods": h["agent_periods"], "chunk_rate": round(h["success_rate"], 4),
                                          "wall_s": round(h["wall_s"], 3)} for h in hist],
                              "stage4_eval_4096_episodes": ev}), flush=True)
# the reference's own stage-4 tables in the same simulator, same evaluation
for mode in a.modes:
    quirks = (Q_PAPER if mode == "paper" else Q_REFERENCE) if a.quirks is None else a.quirks
    ev = {}
    for flavour in ("training", "simulation"):
        h = simulation.evaluate(ROOT / "tests" / "golden" / "assets", 4096, 4, flavour=flavour, quirks=quirks)
        ev[flavour] = {"goal": h["TERMINAL_SUCCESS"] / 4096, "touchdown": h["TERMINAL_CONTACT"] / 4096, "flyzone_x": h["TERMINAL_FLYZONE_X"] / 4096}
    print(json.dumps({"tables": "reference assets (stage 4)", "mode": mode, "stage4_eval_4096_episodes": ev}), flush=True)
