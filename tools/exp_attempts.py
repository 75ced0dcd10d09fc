#!/usr/bin/env python3
"""bench.py's curriculum leg (attempts and all) over other seeds than the bench's twelve: how often a seed needs a second / third curriculum, and where the
worst chosen run ends.   python tools/exp_attempts.py [first_seed] [n_seeds] [accept_touchdown] [max_attempts]  -> one JSON line per seed + a summary"""
import json, sys
from pathlib import Path
from types import SimpleNamespace
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
g.build_hip()
import bench
from dql_multirotor_landing_amd.config import F32
first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 12), (int(sys.argv[2]) if len(sys.argv) > 2 else 36)
if len(sys.argv) > 3:
    bench.CURRICULUM_ACCEPT_TOUCHDOWN = float(sys.argv[3])
attempts = int(sys.argv[4]) if len(sys.argv) > 4 else bench.CURRICULUM_ATTEMPTS
bench.CURRICULUM_SEEDS = tuple(range(first, first + n))
args = SimpleNamespace(curriculum_envs=32768, curriculum_budget=50000, curriculum_seeds=n, curriculum_attempts=attempts)
cur = bench.curriculum_leg(args, None, 1, 0, 0, F32)
if "error" in cur:
    print(json.dumps(cur)); sys.exit(1)
for r in cur["runs"]:
    print(json.dumps({"seed": r["seed"], "attempts": r["attempts"], "accepted": r["accepted"], "promoted_levels": r["promoted_levels"], "wall_to_stage4_s": round(r["wall_to_stage4_s"], 3),
                      "wall_all_levels_s": round(r["wall_all_levels_s"], 3), "chosen": r["stage4_greedy_4096_episodes"], "first_attempt": r["first_attempt"],
                      "selection": [a["selection"]["touchdown_rate"] for a in r["attempt_records"]]}), flush=True)
keep = ("wall_to_stage4_s", "wall_all_levels_s", "seeds_reaching_stage4_by_rule", "n_seeds", "promoted_levels_per_seed", "attempts", "stage4_greedy_4096_episodes")
print(json.dumps({"summary": {k: cur[k] for k in keep}, "accept_touchdown": bench.CURRICULUM_ACCEPT_TOUCHDOWN, "max_attempts": attempts}))
