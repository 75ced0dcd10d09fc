#!/usr/bin/env python3
"""Curriculum experiments in paper mode: promotion threshold x exploration floor x fold semantics."""
import itertools, json, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.trainer import Trainer

grid = [dict(success_rate=0.85, eps_floor=e, fold_per_step=f, n_envs=n)
        for e, f, n in itertools.product([0.0, 0.02, 0.05], [0, 1], [4096])]
for kw in grid:
    with tempfile.TemporaryDirectory() as d:
        tr = Trainer(mode="paper", save_path=Path(d) / "run", chunk_steps=512, max_steps_per_level=60000, max_num_episodes=10**9,
                     checkpoint_every=10**9, **kw)
        hist = tr.curriculum_training()
    print(json.dumps({"cfg": kw, "levels": [(h["level"], h["promoted"], round(h["success_rate"], 3), h["agent_periods"], round(h["wall_since_start_s"], 2)) for h in hist]}), flush=True)
