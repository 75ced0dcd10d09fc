#!/usr/bin/env python3
"""Why does curriculum level 0 plateau?  Trains level 0 only, straight on the Engine (no Trainer), through a scripted exploration
schedule, and prints the terminal histogram of every block of agent periods, then the same tables acting with eps = 0 while still
learning, then acting greedily without learning.  VERDICT r2 item 5: "log the per-code terminal histogram over the last 10 % of the
level's budget at eps = 0.01 vs 0".

    python tools/exp_level0.py [--envs 32768] [--quirks 96] [--random 256] [--decay 1024] [--floor-periods 16384] [--floor 0.01] ...
"""
import argparse, json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from dql_multirotor_landing_amd.config import training_config
from dql_multirotor_landing_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=32768)
ap.add_argument("--quirks", type=int, default=0x60)
ap.add_argument("--level", type=int, default=0)
ap.add_argument("--random", type=int, default=256, help="agent periods at eps = 1")
ap.add_argument("--decay", type=int, default=1024, help="agent periods of linear decay 1 -> floor")
ap.add_argument("--floor-periods", type=int, default=16384)
ap.add_argument("--floor", type=float, default=0.01)
ap.add_argument("--zero-periods", type=int, default=4096, help="then eps = 0, still learning")
ap.add_argument("--eval-periods", type=int, default=2048, help="then greedy, no learning")
ap.add_argument("--block", type=int, default=2048)
ap.add_argument("--ppl", type=int, default=8)
ap.add_argument("--fold-per-step", type=int, default=1)
ap.add_argument("--seed", type=int, default=42)
ap.add_argument("--init-uniform", type=int, default=0)
ap.add_argument("--set", nargs="*", default=[], help="extra DqlConfig fields key=value")
ap.add_argument("--dump", default="")
a = ap.parse_args()
kw = {}
for kv in a.set:
    k, v = kv.split("=")
    kw[k] = float(v) if "." in v or "e" in v else int(v)
cfg = training_config(a.level, quirks=a.quirks, fold_per_step=a.fold_per_step, init_uniform=a.init_uniform, **kw)
eng = Engine(cfg, a.envs, seed=a.seed)
eng.set_option("periods_per_launch", a.ppl)


def block(tag, n, eps_fn, learn=True):
    done = 0
    while done < n:
        m = min(a.block, n - done)
        s0 = eng.stats()
        k = 0
        while k < m:
            c = min(64, m - k)
            if learn:
                eng.train_steps(c, eps_fn(done + k))
            else:
                eng.eval_steps(c)
            k += c
        s1 = eng.stats()
        ep = max(1, s1["episodes"] - s0["episodes"])
        h = {c.replace("TERMINAL_", "").lower(): round((s1["by_code"][c] - s0["by_code"][c]) / ep, 4) for c in s1["by_code"] if s1["by_code"][c] - s0["by_code"][c]}
        print(json.dumps({"phase": tag, "periods": [done, done + m], "eps": [round(eps_fn(done), 4), round(eps_fn(done + m - 1), 4)], "episodes": ep,
                          "steps_per_episode": round((s1["decisions"] - s0["decisions"]) / ep, 1), "hist": h}), flush=True)
        done += m


block("random", a.random, lambda t: 1.0)
block("decay", a.decay, lambda t: 1.0 + (a.floor - 1.0) * t / max(1, a.decay))
block("floor", a.floor_periods, lambda t: a.floor)
tabs_floor = [t.copy() for t in eng.get_tables()]
block("zero", a.zero_periods, lambda t: 0.0)
tabs_zero = [t.copy() for t in eng.get_tables()]
block("eval", a.eval_periods, lambda t: 0.0, learn=False)
# acting vs learning: both table sets, frozen (learning rate 0), flown with eps = 0.01 and eps = 0
train_eng = eng
for name, tabs in (("tables_learnt_at_floor", tabs_floor), ("tables_learnt_at_zero", tabs_zero)):
    for eps in (a.floor, 0.0):
        c2 = training_config(a.level, quirks=a.quirks, fold_per_step=a.fold_per_step, init_uniform=a.init_uniform, alpha_min=0.0, **kw)
        eng = Engine(c2, a.envs, seed=a.seed + 1, alpha_table=np.zeros(1))
        eng.set_option("periods_per_launch", a.ppl)
        eng.set_tables(*tabs)
        eng.train_steps(512, eps)  # flush the first generation of episodes
        block(f"frozen:{name}:eps={eps}", 1024, lambda t, e=eps: e)
        q2 = eng.get_tables()[0]
        assert np.allclose(q2, tabs[0], rtol=1e-9, atol=1e-9), "tables moved under learning rate 0"  # (q - t) + t rounds, nothing more
        eng.close()
eng = train_eng
if a.dump:
    qa, qb, cnt = eng.get_tables()
    np.savez(a.dump, qa=qa, qb=qb, cnt=cnt)
