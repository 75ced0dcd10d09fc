#!/usr/bin/env python3
"""Greedy roll-outs of the reference's stage-4 tables at every working level (training flavour, paper-mode acceleration)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "scripts"))
import simulation
from dql_multirotor_landing_amd.config import Q_PAPER
for level in range(5):
    h = simulation.evaluate(ROOT / "tests" / "golden" / "assets", 4096, level, flavour="training", quirks=Q_PAPER)
    print(json.dumps({"level": level, "goal": h["TERMINAL_SUCCESS"] / 4096, "flyzone_x": h["TERMINAL_FLYZONE_X"] / 4096, "timeout": h["TERMINAL_TIMEOUT"] / 4096}), flush=True)
