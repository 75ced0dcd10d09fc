#!/usr/bin/env python3
"""Experiment: outcomes of greedy roll-outs of the reference's stage-4 tables under simulator variants."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "scripts"))
import simulation
from dql_multirotor_landing_amd.config import Q_REFERENCE, Q_FROZEN_ACC_REFERENCE, Q_STICKY_CHECK
tables = ROOT / "tests" / "golden" / "assets"
variants = {
    "reference": {},
    "acc_finite_difference": {"quirks": Q_REFERENCE & ~Q_FROZEN_ACC_REFERENCE},
    "acc_fd+platform_default_tx1": {"quirks": Q_REFERENCE & ~Q_FROZEN_ACC_REFERENCE, "mp_t_x": 1.0},
    "acc_fd+vz-0.2": {"quirks": Q_REFERENCE & ~Q_FROZEN_ACC_REFERENCE, "vz_setpoint": -0.2},
    "acc_fd+start_within_2m": {"quirks": Q_REFERENCE & ~Q_FROZEN_ACC_REFERENCE, "p_max": 4.5, "init_uniform": 0, "init_sigma": 0.7},
}
for name, kw in variants.items():
    for flavour, level in (("simulation", 4), ("training", 4)):
        h = simulation.evaluate(tables, 4096, level, flavour=flavour, **kw)
        print(json.dumps({"variant": name, "flavour": flavour, "touchdown": h["TERMINAL_CONTACT"] / 4096, "goal": h["TERMINAL_SUCCESS"] / 4096,
                          "flyzone_x": h["TERMINAL_FLYZONE_X"] / 4096, "min_alt": h["TERMINAL_MINIMUM_ALTITUDE"] / 4096, "timeout": h["TERMINAL_TIMEOUT"] / 4096}), flush=True)
