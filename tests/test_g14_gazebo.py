"""G14 — the simulator against the reference's own Gazebo flight records.

The fixture `tests/golden/g14_gazebo_episodes.npz` (written by `tests/golden/make_g14.py` in the build container: a hand-written TFRecord /
protobuf reader over the reference's `assets/**/logs/events.out.tfevents.*`, numbers only) holds one row per episode the reference flew in
Gazebo 11 + ODE + RotorS: episode number, exploration rate, learning rate, success rate, termination code, cumulative and mean reward
(=> episode length).  It is the only Gazebo-side evidence that exists for the simulator rows.

What can be compared without any table: run `22-02-2025 21:37:06` flew its first 801 episodes (`_current_episode` 0..800) at eps = 1.0
(`pkg/trainer.py:112-126`), a uniformly random policy.  This simulator flies the same thing — `mode="reference"` MDP (quirks 0x7f), level 0,
N(0, p_max / 3) start, z_init 4 m, rpm platform, the Trainer's env settings (`pkg/trainer.py:40-43,180`) — 3 * 10^4 .. 10^5 episodes, and every
statistic of it has to fall inside the 99 % interval (Wilson / bootstrap) of the 801 Gazebo samples.

What came out: the statistics agree — termination mix, episode-length mean and median overall and per outcome,
cumulative reward (which counts the sticky-success steps, B7 / B8), the share of goal episodes that take the minimum 23 steps — WITH THE
PARAMETERS THE MANAGER NODE ACTUALLY RESOLVED (`config.as_launched_config`: the launch file's private parameters never reach the code, which
falls back to t_x = 1 m/s and observation noise 0.25 m / 0.1 m/s), and disagree grossly with the launch file's literal values (t_x = 1.6,
noise 0: goal 27 % against Gazebo's 38 %).  The second test below keeps that discrimination.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from dql_multirotor_landing_amd.config import CHECK_NAMES, DqlConfig, F32, F64, as_launched_config  # noqa: E402

G14 = ROOT / "tests" / "golden" / "g14_gazebo_episodes.npz"
GOAL, FLY_X = CHECK_NAMES.index("TERMINAL_SUCCESS"), CHECK_NAMES.index("TERMINAL_FLYZONE_X")
MAIN_RUN = "22-02-2025 21:37:06"


def gazebo_run(name=MAIN_RUN):
    d = np.load(G14)
    r = list(d["runs"]).index(name)
    m = d["run"] == r
    return {k: d[k][m] for k in d.files if k not in ("runs", "run_duplicate_of")}


def gazebo_random_policy_block():
    g = gazebo_run()
    m = g["exploration_rate"] == 1.0
    return g["code"][m].astype(int), g["steps"][m].astype(np.float64), g["cumulative_reward"][m].astype(np.float64)


def wilson(k, n, z=2.576):
    p = k / n
    c = (p + z * z / (2 * n)) / (1 + z * z / n)
    h = z * np.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / (1 + z * z / n)
    return (0.0 if k == 0 else c - h), (1.0 if k == n else c + h)


def bootstrap(x, f, n=4000, seed=0):
    rng = np.random.RandomState(seed)
    idx = rng.randint(0, len(x), (n, len(x)))
    v = np.array([f(x[i]) for i in idx])
    return tuple(np.percentile(v, [0.5, 99.5]))


def fly_random_policy(make_engine, cfg, n_envs, periods, read, min_generations=8):
    """eps = 1 training periods; every finished episode's (code, steps, cumulative reward, generation).  Only the generations ALL envs
    completed are kept: a fixed horizon would otherwise favour short episodes."""
    eng = make_engine(cfg, n_envs)
    gen = np.zeros(n_envs, int)
    rows = []
    for _ in range(periods):
        eng.train_steps(1, 1.0)
        done, code, steps, cum = read(eng)
        if done.any():
            rows.append(np.stack([code[done], steps[done], cum[done], gen[done]]))
            gen[done] += 1
    code, steps, cum, g = np.concatenate(rows, axis=1)
    k = g < gen.min()
    assert gen.min() >= min_generations, gen.min()
    return code[k].astype(int), steps[k], cum[k]


def read_oracle(o):
    if not hasattr(o, "_g14_idx"):
        rn, ino = o.field_names(False), o.field_names(True)
        o._g14_idx = (rn.index("cum_x"), rn.index("reward"), ino.index("step_count"), ino.index("code"), ino.index("flags"))
    r, i = o.get_fields()
    ic, ir, isc, icode, ifl = o._g14_idx
    return (i[ifl] & 1) != 0, i[icode], i[isc].astype(np.float64), r[ic] - r[ir]  # the reference logs the sum BEFORE the terminal step's reward


def read_engine(e):
    so = e.step_outputs_view()
    return so["done"] != 0, so["code"].astype(int), so["step_count"].astype(np.float64), so["cumulative_reward"] - so["reward"]


def compare_with_gazebo(code, steps, cum, expect_agreement=True):
    """-> dict statistic: (ours, (lo, hi) of the 801 Gazebo samples, inside?)"""
    gc, gs, gcum = gazebo_random_policy_block()
    n = len(gc)
    out = {}

    def put(name, ours, ci):
        out[name] = (float(ours), (float(ci[0]), float(ci[1])), bool(ci[0] <= ours <= ci[1]))

    put("goal fraction", (code == GOAL).mean(), wilson((gc == GOAL).sum(), n))
    put("fly-zone x fraction", (code == FLY_X).mean(), wilson((gc == FLY_X).sum(), n))
    put("other terminations", ((code != GOAL) & (code != FLY_X)).mean(), wilson(((gc != GOAL) & (gc != FLY_X)).sum(), n))
    put("steps mean", steps.mean(), bootstrap(gs, np.mean))
    put("steps median", np.median(steps), bootstrap(gs, np.median))
    put("cumulative reward mean", cum.mean(), bootstrap(gcum, np.mean))
    for c, tag in ((GOAL, "goal"), (FLY_X, "fly-zone x")):
        k, gk = code == c, gc == c
        put(f"{tag}: steps mean", steps[k].mean(), bootstrap(gs[gk], np.mean))
        put(f"{tag}: steps median", np.median(steps[k]), bootstrap(gs[gk], np.median))
        put(f"{tag}: cumulative reward mean", cum[k].mean(), bootstrap(gcum[gk], np.mean))
    put("goal episodes of the minimum 23 steps", (steps[code == GOAL] == 23).mean(), bootstrap((gs[gc == GOAL] == 23).astype(float), np.mean))
    return out


# ---------------------------------------------------------------- the fixture itself (CPU)
def test_g14_fixture_is_the_reference_runs_record():
    d = np.load(G14)
    runs = list(d["runs"])
    assert runs[1] == MAIN_RUN and runs[2] == "x_direction/" + MAIN_RUN and d["run_duplicate_of"][2] == 1  # the copy is listed, not stored
    g = gazebo_run()
    n = len(g["episode"])
    assert n == 19538 and np.array_equal(g["episode"], np.arange(1, n + 1))  # one event file per episode, step = _curriculum_episode_count
    assert (g["code"] >= 0).all()  # every termination text is a CheckResult value
    # the exploration schedule of pkg/trainer.py:112-126 on _current_episode = episode - 1 at level 0; 0 after the promotion
    first_l1 = int(np.argmax(g["success_rate"] > 0.96)) + 1  # index of the first level-1 episode
    ep0 = np.arange(first_l1, dtype=np.float64)
    eps = np.where(ep0 <= 800, 1.0, np.maximum(1 + (0.01 - 1) * (ep0 - 800) / 1200.0, 0.01))
    np.testing.assert_allclose(g["exploration_rate"][:first_l1], eps, rtol=1e-6)
    assert (g["exploration_rate"][first_l1:] == 0).all()
    # "Success rate" = sum(deque(maxlen=100) of goal flags) / 100, deque cleared at the promotion (pkg/trainer.py:218-232): the
    # termination codes recovered from the text records reproduce the logged scalar for all 19 538 episodes
    goal = (g["code"] == GOAL).astype(int)
    want = np.empty(n)
    for seg in (slice(0, first_l1), slice(first_l1, n)):
        c = np.concatenate([[0], np.cumsum(goal[seg])])
        i = np.arange(1, len(c))
        want[seg] = (c[i] - c[np.maximum(i - 100, 0)]) / 100.0
    np.testing.assert_allclose(g["success_rate"], want, atol=1e-6)
    assert first_l1 == 18282 and g["success_rate"][first_l1 - 1] == np.float32(0.97)
    # episode length: cumulative / mean reward is an integer to rounding, within the MDP's limits (time-out at step 459, B18)
    # (check() writes both before reward() adds the terminal step's reward: an episode that ends at its first step logs 0 and 0)
    k = g["mean_reward"] != 0
    ratio = g["cumulative_reward"][k].astype(np.float64) / g["mean_reward"][k].astype(np.float64)
    assert np.abs(ratio - g["steps"][k]).max() < 0.02 and (g["steps"][~k] == 1).all() and (~k).sum() == 55
    assert g["steps"].min() >= 1 and g["steps"].max() <= 459
    assert set(np.unique(g["steps"][g["code"] == CHECK_NAMES.index("TERMINAL_TIMEOUT")])) == {459}
    # the random-policy block the tests below compare with
    gc, gs, gcum = gazebo_random_policy_block()
    assert len(gc) == 801 and (gc == GOAL).sum() == 304 and (gc == FLY_X).sum() == 497
    assert gs[gc == GOAL].min() == 23  # f_ag = 22.92 goal-bin steps: 23 (pkg/mdp.py:418)


def _oracle(cfg, n):
    from oracle.oracle import Oracle
    return Oracle(cfg, n, seed=42, n_threads=4)


def _report(res):
    return "\n".join(f"  {k:45s} ours {v[0]:9.4f}   Gazebo 99 % [{v[1][0]:9.4f}, {v[1][1]:9.4f}]  {'ok' if v[2] else 'OUTSIDE'}" for k, v in res.items())


def test_g14_random_policy_statistics_oracle():
    """CPU oracle, float64, 2 048 envs: >= 8 complete generations = >= 16 384 episodes"""
    code, steps, cum = fly_random_policy(_oracle, as_launched_config(dtype=F64), 2048, 1100, read_oracle)
    assert len(code) >= 16384
    res = compare_with_gazebo(code, steps, cum)
    print(_report(res))
    assert all(v[2] for v in res.values()), "\n" + _report(res)


def test_g14_discriminates_the_launch_files_literal_values():
    """the launch file's t_x = 1.6 m/s, noise 0 (`DqlConfig()` defaults; what round 1-3 flew) is NOT what Gazebo flew: the platform is faster
    than the velocity goal bin half of the time, so far fewer episodes start inside the goal state and collect its sticky reward"""
    code, steps, cum = fly_random_policy(_oracle, DqlConfig(dtype=F64), 2048, 1100, read_oracle, min_generations=4)
    res = compare_with_gazebo(code, steps, cum)
    print(_report(res))
    assert not res["goal fraction"][2] and res["goal fraction"][0] < 0.30
    assert not res["cumulative reward mean"][2] and res["cumulative reward mean"][0] < 0
    assert not res["goal episodes of the minimum 23 steps"][2]
    # what does not depend on the platform's speed agrees either way: how long a random pitch walk takes to leave the fly zone
    assert res["fly-zone x: steps median"][2]


# ---------------------------------------------------------------- the HIP path (MI355X)
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [F32, F64])
def test_g14_random_policy_statistics_hip(dtype):
    """the product path through the C ABI: 16 384 envs, >= 10^5 episodes, the dtype every throughput figure is measured in and float64"""
    from dql_multirotor_landing_amd.engine import Engine

    def make(cfg, n):
        return Engine(cfg, n, seed=42)

    code, steps, cum = fly_random_policy(make, as_launched_config(dtype=dtype), 16384, 1500, read_engine)
    assert len(code) >= 100000
    res = compare_with_gazebo(code, steps, cum)
    print(_report(res))
    assert all(v[2] for v in res.values()), "\n" + _report(res)


def test_g14_sequential_reference_learner_first_three_thousand_episodes():
    """G14, second cut, in short: the reference's OWN one-env algorithm (`guess` -> `env.step` -> `update` with alpha(pre-increment count), its eps
    schedule, all quirks) on the oracle's one-env step with external actions + the oracle's sequential `agent_update` — the path the single-env
    drop-in classes take — for the first 3 000 episodes of level 0 (eps = 1, the decay, and the first thousand at the floor 0.01), one seed, ~12 s.  The eps = 1 block (episodes 0 .. 800) must again sit inside
    the Gazebo sample's 99 % interval (a different code path from the batched random-policy test above), and the eps-decay block (episodes
    1 000 .. 2 000) near the Gazebo run's 0.538: 0.55 .. 0.60 over 8 seeds (profiles/r4_g14_learning_curves.jsonl, 19 000 episodes per seed and on
    through the B6 collapse at level 1); a loose band here, the learner is seed-noisy."""
    sys.path.insert(0, str(ROOT / "tests"))
    import g14_learning_curve as lc
    codes, steps, _, _ = lc.run(3, as_launched_config(dtype=F64), 3000)
    gc, gs, _ = gazebo_random_policy_block()
    lo, hi = wilson((gc == GOAL).sum(), len(gc))
    p0 = (codes[:801] == GOAL).mean()
    assert lo <= p0 <= hi, (p0, lo, hi)
    g = gazebo_run()
    gaz = (g["code"][1000:2000] == GOAL).mean()
    p1 = (codes[1000:2000] == GOAL).mean()
    assert 0.5 < gaz < 0.58 and abs(p1 - gaz) < 0.12, (p1, gaz)
    assert abs(steps[:801].mean() - gs.mean()) < 8.0
    # round 5: the block flown at the schedule's floor eps = 0.01 (episodes 2 000 .. 3 000; pkg/trainer.py:112-126) — what the learner
    # has LEARNT by then, acting almost greedily under the reference's quirks.  Gazebo: goal share 0.534, mean length 46.1 steps; this simulator
    # 0.51 / 0.62 / 0.65 and 48.5 / 40.8 / 50.0 steps for seeds 1 / 2 / 3.  The band is the seed spread, not a rounding tolerance.
    gaz2, gaz2_steps = (g["code"][2000:3000] == GOAL).mean(), g["steps"][2000:3000].mean()
    p2 = (codes[2000:3000] == GOAL).mean()
    assert 0.5 < gaz2 < 0.57 and abs(p2 - gaz2) < 0.15, (p2, gaz2)
    assert abs(steps[2000:3000].mean() - gaz2_steps) < 10.0, (steps[2000:3000].mean(), gaz2_steps)
    assert p2 > p0 + 0.08   # it learnt: clearly above the random policy's share, as the Gazebo run was (0.534 vs 0.380)
