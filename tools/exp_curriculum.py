#!/usr/bin/env python3
"""Curriculum experiments: one Trainer run per (seed, keyword set); per level: promoted?, episodes, online outcome mix over the
level's last quarter (from logs/scalars.csv), then greedy evaluation of the stage-4 tables (goal-hold in the training flavour,
touchdown in the landing flavour).

    python tools/exp_curriculum.py --seeds 42 1 2 --set name:key=value,key=value ... [--envs 4096] [--budget-per-env 64]
"""
import argparse, csv, json, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "scripts"))
import simulation
from dql_multirotor_landing_amd.config import Q_PAPER
from dql_multirotor_landing_amd.trainer import Trainer


def parse_val(v):
    for f in (int, float):
        try:
            return f(v)
        except ValueError:
            pass
    return {"None": None, "True": True, "False": False}.get(v, v)


ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--budget-per-env", type=float, default=64)
ap.add_argument("--seeds", type=int, nargs="+", default=[42, 1, 2])
ap.add_argument("--set", nargs="+", default=["default:"], help="name:key=value,key=value (Trainer keywords; ppl = periods_per_launch)")
ap.add_argument("--eval-envs", type=int, default=4096)
a = ap.parse_args()
for spec in a.set:
    name, _, kvs = spec.partition(":")
    kw = {k: parse_val(v) for k, v in (kv.split("=") for kv in kvs.split(",") if kv)}
    if kw.pop("as_launched", 0):  # the world the reference's manager node resolved under roslaunch (config.AS_LAUNCHED, golden G14)
        from dql_multirotor_landing_amd.config import AS_LAUNCHED
        kw["env_kw"] = dict(AS_LAUNCHED)
    if "ppl" in kw:
        kw["periods_per_launch"] = kw.pop("ppl")
    budget_per_env = kw.pop("budget_per_env", a.budget_per_env)
    envs = kw.pop("envs", a.envs)
    for seed in a.seeds:
        with tempfile.TemporaryDirectory() as d:
            t0 = time.perf_counter()
            tr = Trainer(mode="paper", n_envs=envs, save_path=Path(d) / "run", chunk_steps=64, max_num_episodes=int(budget_per_env * envs), checkpoint_every=10**9,
                         seed=seed, **kw)
            hist = tr.curriculum_training()
            wall = time.perf_counter() - t0
            rows = list(csv.DictReader(open(Path(d) / "run" / "logs" / "scalars.csv")))
            levels = []
            for h in hist:
                r = [x for x in rows if int(x["Curriculum step"]) == h["level"]]
                tail = r[-max(1, len(r) // 4):]
                tot = {c: sum(float(x[f"Episode/Termination Condition/{c}"]) for x in tail) for c in
                       ("TERMINAL_SUCCESS", "TERMINAL_FLYZONE_X", "TERMINAL_TIMEOUT", "TERMINAL_MINIMUM_ALTITUDE", "TERMINAL_FLYZONE_Z", "TERMINAL_CONTACT")}
                n_ep = max(1.0, sum(tot.values()))
                levels.append({"level": h["level"], "promoted": h["promoted"], "pop": None if h["success_rate"] is None else round(h["success_rate"], 3), "episodes_per_env": round(h["episodes"] / envs, 1), "periods": h["agent_periods"],
                               "restarts": h.get("restarts", 0), "step_backs": h.get("step_backs", 0), "tail": {k.replace("TERMINAL_", "").lower(): round(v / n_ep, 3) for k, v in tot.items() if v}})
            wk = kw.get("env_kw") or {}  # evaluated in the world it was trained in
            ev_t = simulation.evaluate(Path(d) / "run", a.eval_envs, 4, flavour="training", quirks=Q_PAPER, **wk)
            ev_s = simulation.evaluate(Path(d) / "run", a.eval_envs, 4, flavour="simulation", quirks=Q_PAPER, **wk)
            tr._engine.close()
        print(json.dumps({"set": name, "kw": kw, "envs": envs, "budget_per_env": budget_per_env, "seed": seed, "wall_s": round(wall, 2),
                          "wall_to_stage4_s": round(hist[3].get("wall_first_promoted_s") or hist[3]["wall_since_start_s"], 2) if len(hist) > 3 else None,
                          "promoted_levels": sum(1 for h in hist if h["promoted"]), "goal_hold": round(ev_t["TERMINAL_SUCCESS"] / a.eval_envs, 3),
                          "touchdown": round(ev_s["TERMINAL_CONTACT"] / a.eval_envs, 3), "flyzone_landing": round(ev_s["TERMINAL_FLYZONE_X"] / a.eval_envs, 3),
                          "levels": levels}), flush=True)
