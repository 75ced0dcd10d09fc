#!/usr/bin/env python3
"""Which of the reference's quirks (include/dql.h DQL_Q_*) keep the curriculum from learning in this simulator?
bench.py's recipe (quirks 0x60 = the reference's update rule B1/B2 + its goal counter) with ONE more quirk bit set at a time, then all
of them (0x7f = mode "reference" except for when the level transfer is applied), a few seeds each: levels promoted by the rule and the
greedy stage-4 roll-outs bench.py reports.   python tools/exp_quirk_ablation.py [envs=32768] [seeds=3]"""
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
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NAMES = {0x01: "B7 fail term every step", 0x02: "B8 sticky check result", 0x04: "B9 shaping survives reset", 0x08: "B19 frozen acceleration reference",
         0x10: "B3 bootstrap only on position-bin change"}
only = [int(x, 0) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (0x00, 0x01, 0x02, 0x04, 0x08, 0x10, 0x1f)
for extra in only:
    q = 0x60 | extra
    for seed in bench.CURRICULUM_SEEDS[:n_seeds]:
        kw = dict(bench.CURRICULUM_KW, quirks=q)
        with tempfile.TemporaryDirectory() as d:
            tr = Trainer(mode="paper", n_envs=n, dtype=F32, save_path=Path(d) / "run", chunk_steps=64, sync_period=bench.CURRICULUM_SYNC, max_num_episodes=384 * n,
                         checkpoint_every=10**9, seed=seed, **kw)
            t0 = time.perf_counter(); h = tr.curriculum_training(); wall = time.perf_counter() - t0
            ev = simulation.evaluate(Path(d) / "run", 4096, 4, flavour="training", quirks=Q_PAPER)
            td = simulation.evaluate(Path(d) / "run", 4096, 4, flavour="simulation", quirks=Q_PAPER)
            # observation quirks change what the state MEANS: a policy trained under B19 is also flown under B19
            same = {}
            if extra & 0x08:
                same = {"goal_hold_rate_same_observation_quirk": simulation.evaluate(Path(d) / "run", 4096, 4, flavour="training", quirks=Q_PAPER | 0x08)["TERMINAL_SUCCESS"] / 4096,
                        "touchdown_rate_same_observation_quirk": simulation.evaluate(Path(d) / "run", 4096, 4, flavour="simulation", quirks=Q_PAPER | 0x08)["TERMINAL_CONTACT"] / 4096}
            tr._engine.close()
        print(json.dumps({"quirks": hex(q), "added": NAMES.get(extra, "none" if not extra else "all five (the reference's full set)"), "seed": seed, "envs": n,
                          "promoted": [bool(x["promoted"]) for x in h], "online_success_at_handover": [round(x["success_rate"], 3) for x in h], "wall_s": round(wall, 2),
                          "goal_hold_rate": ev["TERMINAL_SUCCESS"] / 4096, "touchdown_rate": td["TERMINAL_CONTACT"] / 4096, **same}), flush=True)
