#!/usr/bin/env python3
"""G14, second cut: the reference's OWN algorithm, one env at a time, in this simulator — against the learning curve of its Gazebo run.

Not a pytest module (a minute per seed): an experiment whose output is committed under profiles/ (r4_g14_learning_curves.jsonl).  Lives under
tests/ because it drives the CPU oracle.

    python tests/g14_learning_curve.py SEED [launched|file] [EPISODES] [LEVEL-1 EPISODES after the promotion, e.g. 1256 as in the Gazebo run]

The reference trains ONE env sequentially (`pkg/trainer.py:187-236`): `guess` (eps schedule of :112-126, `np.random` uniform + randint always
drawn, B4) -> `env.step` -> `update` with alpha(pre-increment count) (B5), all quirks of SURVEY.md appendix B (quirks 0x7f: B1-B3 update
rule, B7 / B8 reward, B9 shaping, B19 acceleration).  Its run `22-02-2025 21:37:06` (golden G14) needed 18 282 episodes for the first
97 / 100 window at level 0 and hovered at 53-76 % goal episodes per thousand for most of them.  Here the same loop runs on the oracle's one-env
step (float64) + the oracle's sequential `agent_update`, with the parameters the manager node resolved under roslaunch ("launched":
`config.as_launched_config`) or the launch file's literal values ("file").  Output: goal share and mean length per 1 000 episodes, first
episode at which the deque rule fires, next to the Gazebo run's curve."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from dql_multirotor_landing_amd.config import CHECK_NAMES, DqlConfig, F64, N_CELLS, as_launched_config  # noqa: E402
from oracle.oracle import Oracle, agent_update  # noqa: E402

GOAL = CHECK_NAMES.index("TERMINAL_SUCCESS")


def eps_of(ep):  # pkg/trainer.py:112-126 at level 0
    return 1.0 if ep <= 800 else max(1 + (0.01 - 1) * (ep - 800) / 1200.0, 0.01)


def run(seed, cfg, episodes, level1_episodes=0):
    """-> (codes, steps) of level 0, and — when level1_episodes > 0 and the deque rule fired — of the level-1 episodes flown afterwards as the reference
    flies them: `transfer_learning(0, 1.0)` first (B6: Q[0] = Q[-1] = zeros), then a new env at level 1 (uniform starts, eps = 0)"""
    from oracle.oracle import transfer
    o = Oracle(cfg, 1, seed=seed)
    rng = np.random.RandomState(seed)
    rn, ino = o.field_names(False), o.field_names(True)
    i_rew, i_idx, i_fl, i_code, i_sc = rn.index("reward"), ino.index("idx_x"), ino.index("flags"), ino.index("code"), ino.index("step_count")
    qa, qb, cnt = np.zeros(N_CELLS), np.zeros(N_CELLS), np.zeros(N_CELLS)
    atab = cfg.alpha_table()
    act = np.zeros(1, np.uint8)
    o.step(act)  # the env is created flagged for reset: this period places it and returns the first state (TrainingLandingEnv.reset)
    s = int(o.get_fields()[1][i_idx][0])
    codes, steps, ep = [], [], 0
    codes1, steps1 = [], []
    sa, nsa, al, rw = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1), np.zeros(1)
    window = []
    level = 0
    while True:
        eps = eps_of(ep) if level == 0 else 0.0
        u, ra = rng.uniform(), rng.randint(3)
        a = ra if u < eps else int(np.argmax((qa[3 * s:3 * s + 3] + qb[3 * s:3 * s + 3]) / 2))
        act[0] = a
        o.step(act)
        r, i = o.get_fields()
        c = int(cnt[3 * s + a])
        sa[0], nsa[0], al[0], rw[0] = 3 * s + a, i[i_idx][0], (atab[c] if c < len(atab) else cfg.alpha_min), r[i_rew][0]
        agent_update(qa, qb, cnt, sa, nsa, al, cfg.gamma, rw, quirks=cfg.quirks)
        if i[i_fl][0] & 1:
            code = int(i[i_code][0])
            (codes if level == 0 else codes1).append(code); (steps if level == 0 else steps1).append(int(i[i_sc][0])); ep += 1
            if level == 0:
                window = (window + [int(code == GOAL)])[-100:]
                if level1_episodes > 0 and sum(window) > 96:  # pkg/trainer.py:232-243: promotion, transfer of the FINISHED level (B6), next level's env
                    transfer(qa, qb, 0, 1.0)
                    o.set_curriculum(1)
                    level, ep = 1, 0
                elif ep >= episodes:
                    break
            elif ep >= level1_episodes:
                break
            o.step(act)
            s = int(o.get_fields()[1][i_idx][0])
        else:
            s = int(nsa[0])
    return np.array(codes), np.array(steps), np.array(codes1), np.array(steps1)


def curve(codes, steps):
    goal = (codes == GOAL).astype(int)
    w = np.convolve(goal, np.ones(100, int))[:len(goal)]
    first = np.where(w > 96)[0]
    return {"goal_share_per_1000_episodes": [round(float(goal[a:a + 1000].mean()), 3) for a in range(0, len(goal) - 999, 1000)],
            "mean_steps_per_1000_episodes": [round(float(steps[a:a + 1000].mean()), 1) for a in range(0, len(goal) - 999, 1000)],
            "first_promotion_episode": int(first[0]) + 1 if len(first) else None}


if __name__ == "__main__":
    if sys.argv[1] == "gazebo":
        d = np.load(ROOT / "tests" / "golden" / "g14_gazebo_episodes.npz")
        m = d["run"] == 1
        n0 = 18282  # level 0 of the run (tests/test_g14_gazebo.py)
        c1, s1 = d["code"][m][n0:], d["steps"][m][n0:]
        print(json.dumps({"what": "reference + Gazebo, run 22-02-2025 21:37:06, level 0 (golden G14)", **curve(d["code"][m][:n0], d["steps"][m][:n0]),
                          "level1": {"episodes": int(len(c1)), "mix": {CHECK_NAMES[k]: round(float((c1 == k).mean()), 4) for k in np.unique(c1)},
                                     "steps_mean": round(float(s1.mean()), 1), "steps_median": float(np.median(s1))}}))
        sys.exit(0)
    seed = int(sys.argv[1])
    which = sys.argv[2] if len(sys.argv) > 2 else "launched"
    episodes = int(sys.argv[3]) if len(sys.argv) > 3 else 19000
    cfg = as_launched_config(dtype=F64) if which == "launched" else DqlConfig(dtype=F64)
    t0 = time.time()
    l1 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    codes, steps, codes1, steps1 = run(seed, cfg, episodes, l1)
    out = {"what": f"this simulator (oracle, float64, N = 1, sequential reference algorithm, quirks 0x7f), parameters: {which}", "seed": seed, **curve(codes, steps)}
    if l1 and len(codes1):
        out["level1"] = {"episodes": int(len(codes1)), "mix": {CHECK_NAMES[k]: round(float((codes1 == k).mean()), 4) for k in np.unique(codes1)},
                         "steps_mean": round(float(steps1.mean()), 1), "steps_median": float(np.median(steps1))}
    out["wall_s"] = round(time.time() - t0, 1)
    print(json.dumps(out), flush=True)
