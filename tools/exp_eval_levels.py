#!/usr/bin/env python3
"""Greedy roll-outs at every working level (training flavour, paper mode): the reference's stage-4 tables next to tables trained
here (one curriculum run, saved under gpurun_out/trained_tables)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "scripts"))
import simulation
from dql_multirotor_landing_amd.config import Q_PAPER
from dql_multirotor_landing_amd.trainer import Trainer
envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 42
out = ROOT / "gpurun_out" / "trained_tables"
tr = Trainer(mode="paper", n_envs=envs, save_path=out, chunk_steps=64, max_num_episodes=budget, checkpoint_every=10**9, seed=seed)
hist = tr.curriculum_training()
print(json.dumps({"trained": {"envs": envs, "budget": budget, "seed": seed, "chunk_rates": [round(h["success_rate"], 3) for h in hist]}}), flush=True)
for name, d in (("reference", ROOT / "tests" / "golden" / "assets"), ("trained", out)):
    for level in range(5):
        h = simulation.evaluate(d, 4096, level, flavour="training", quirks=Q_PAPER)
        print(json.dumps({"tables": name, "level": level, "goal": h["TERMINAL_SUCCESS"] / 4096, "flyzone_x": h["TERMINAL_FLYZONE_X"] / 4096, "timeout": h["TERMINAL_TIMEOUT"] / 4096}), flush=True)
    h = simulation.evaluate(d, 4096, 4, flavour="simulation", quirks=Q_PAPER)
    print(json.dumps({"tables": name, "flavour": "simulation", "touchdown": h["TERMINAL_CONTACT"] / 4096, "flyzone_x": h["TERMINAL_FLYZONE_X"] / 4096, "min_alt": h["TERMINAL_MINIMUM_ALTITUDE"] / 4096}), flush=True)
