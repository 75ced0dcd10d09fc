"""Several whole curricula, the first good one kept (dql_multirotor_landing_amd/attempts.py): the selection logic on scripted trainers (CPU),
and the rank-0 score reaching every rank through the control plane's all-reduce."""
import numpy as np
import pytest

from dql_multirotor_landing_amd.attempts import SEED_STRIDE, attempt_seed, curriculum_attempts


class ScriptedTrainer:
    def __init__(self, j, promoted, touchdown, stage4_wall=0.5):
        self.j, self.touchdown, self.closed = j, touchdown, 0
        self.hist = [{"level": k, "promoted": bool(p), "wall_since_start_s": 0.1 * (k + 1), "wall_first_promoted_s": (stage4_wall if k == 3 and p else None)}
                     for k, p in enumerate(promoted)]

    def curriculum_training(self):
        return self.hist


def run(script, **kw):
    made = []

    def make(j):
        made.append(ScriptedTrainer(j, *script[j]))
        return made[-1]

    def close(tr):
        tr.closed += 1

    res = curriculum_attempts(make, lambda tr: {"touchdown_rate": tr.touchdown, "goal_hold_rate": 0.95}, close=close, **kw)
    return res, made


def test_first_attempt_with_every_level_by_the_rule_and_a_good_landing_is_taken():
    res, made = run([([1, 1, 1, 1, 1], 0.82), ([1, 1, 1, 1, 0], 0.93), ([1, 1, 1, 1, 1], 0.88), ([1, 1, 1, 1, 1], 0.99)], max_attempts=6, accept_touchdown=0.87)
    assert res["chosen"] == 2 and res["accepted"] and len(res["attempts"]) == 3 and len(made) == 3  # (the fourth is never trained)
    assert res["trainer"] is made[2] and res["history"] is made[2].hist
    assert [t.closed for t in made] == [1, 1, 1]
    a = res["attempts"]
    assert [x["promoted_levels"] for x in a] == [5, 4, 5] and [x["all_levels_by_rule"] for x in a] == [True, False, True]
    assert a[1]["selection"] == {"touchdown_rate": 0.93, "goal_hold_rate": 0.95}
    # every attempt entered its last level with the levels before it promoted: the first one's time on the call's clock is its own
    assert a[0]["wall_last_level_by_rule_s"] == pytest.approx(0.5, abs=0.05) and a[0]["wall_last_level_by_rule_s"] < a[2]["wall_last_level_by_rule_s"] + 1.0


def test_nothing_accepted_keeps_most_levels_then_best_landing_then_earliest():
    res, made = run([([1, 1, 1, 1, 0], 0.95), ([1, 1, 1, 1, 1], 0.80), ([1, 1, 1, 1, 1], 0.84), ([1, 1, 1, 1, 1], 0.84)], max_attempts=4, accept_touchdown=0.87)
    assert not res["accepted"] and res["chosen"] == 2 and len(res["attempts"]) == 4 and res["trainer"] is made[2]


def test_a_level_handed_over_below_the_last_has_no_stage_by_rule_time():
    res, _ = run([([1, 1, 0, 1, 1], 0.9)], max_attempts=1)
    assert not res["accepted"] and res["attempts"][0]["wall_last_level_by_rule_s"] is None and res["attempts"][0]["promoted_levels"] == 4


def test_one_attempt_is_the_plain_run_and_zero_is_refused():
    res, made = run([([1, 1, 1, 1, 1], 0.5)], max_attempts=1)
    assert res["chosen"] == 0 and not res["accepted"] and len(made) == 1
    with pytest.raises(ValueError):
        run([], max_attempts=0)


def test_attempt_seeds_are_distinct_and_attempt_zero_is_the_seed_itself():
    assert attempt_seed(42, 0) == 42 and attempt_seed(42, 3) == 42 + 3 * SEED_STRIDE
    assert len({attempt_seed(s, j) for s in range(12) for j in range(6)} | {42}) == 12 * 6 + 1


def test_rank_zero_scores_and_every_rank_decides_alike():
    """two ranks in turn against a comm whose all-reduce adds what the other rank contributed: rank 1 never calls `score`"""
    class Comm:
        def __init__(self, other):
            self.other = np.asarray(other, dtype=np.float64)

        def all_reduce_sum(self, v):
            return np.asarray(v, dtype=np.float64) + self.other

    script = [([1, 1, 1, 1, 1], 0.80), ([1, 1, 1, 1, 1], 0.90)]
    scored = []

    def drive(rank, comm_for_attempt):
        k = [0]

        class C:
            def all_reduce_sum(self, v):
                out = comm_for_attempt[k[0]].all_reduce_sum(v); k[0] += 1
                return out

        def score(tr):
            scored.append(rank)
            return {"touchdown_rate": tr.touchdown, "goal_hold_rate": 0.95}
        return curriculum_attempts(lambda j: ScriptedTrainer(j, *script[j]), score, max_attempts=3, accept_touchdown=0.87, comm=C(), rank=rank, close=lambda tr: None)

    r0 = drive(0, [Comm([0, 0]), Comm([0, 0])])                 # rank 1 contributes zeros
    r1 = drive(1, [Comm([0.80, 0.95]), Comm([0.90, 0.95])])    # rank 0 contributes its score
    assert scored == [0, 0]
    assert r0["chosen"] == r1["chosen"] == 1 and r0["accepted"] and r1["accepted"]
    assert [a["selection"] for a in r0["attempts"]] == [a["selection"] for a in r1["attempts"]]


@pytest.mark.gpu
def test_landing_score_is_what_the_evaluation_script_reports(golden_dir):
    """evaluation.landing_score (what attempts are selected with, what bench.py reports) == the two histograms scripts/simulation.py's harness returns"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))
    import simulation
    from dql_multirotor_landing_amd.config import Q_PAPER
    from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
    from dql_multirotor_landing_amd.evaluation import first_episode_outcomes, landing_score
    tables = DoubleQLearningAgent.load(golden_dir / "assets")._padded()
    sc = landing_score(tables, 512, 4, seed=123)
    h = simulation.evaluate(golden_dir / "assets", 512, 4, flavour="simulation", quirks=Q_PAPER)
    g = simulation.evaluate(golden_dir / "assets", 512, 4, flavour="training", quirks=Q_PAPER)
    assert sc == {"touchdown_rate": h["TERMINAL_CONTACT"] / 512, "goal_hold_rate": g["TERMINAL_SUCCESS"] / 512}
    assert 0.8 < sc["touchdown_rate"] < 0.95 and 0.9 < sc["goal_hold_rate"] <= 1.0   # (the reference's stage-4 tables: 0.876 / 0.954 over 4 096)
    assert first_episode_outcomes(tables, 512, 4, seed=977) != h                      # another batch, another draw
    with pytest.raises(ValueError):
        first_episode_outcomes(tables, 8, 4, flavour="gazebo")


@pytest.mark.gpu
def test_attempts_on_the_hip_engine_are_repeatable_and_differ_by_seed(tmp_path):
    """two real (short) curricula per call, none acceptable (threshold above 1): both are trained and scored, the better one is kept; the same call again chooses
    the same attempt with identical tables; attempt 0 is the plain Trainer run of the seed"""
    from dql_multirotor_landing_amd.attempts import SELECTION_SEED
    from dql_multirotor_landing_amd.evaluation import landing_score
    from dql_multirotor_landing_amd.trainer import Trainer
    kw = dict(curriculum_steps=3, n_envs=512, chunk_steps=16, checkpoint_every=10**9, max_num_episodes=4000, t_max=4, mode="paper", judge_envs=64,
              successive_successful_episodes=20, success_rate=0.5, periods_per_launch=4)

    def call(tag):
        def make(j):
            return Trainer(save_path=tmp_path / f"{tag}{j}", seed=attempt_seed(5, j), **kw)

        def score(tr):
            return landing_score(tr._double_q_learning_agent._padded(), 256, 2, seed=SELECTION_SEED)
        return curriculum_attempts(make, score, max_attempts=2, accept_touchdown=1.5)

    a, b = call("a"), call("b")
    assert len(a["attempts"]) == 2 and not a["accepted"] and a["chosen"] == b["chosen"]
    assert [x["selection"] for x in a["attempts"]] == [x["selection"] for x in b["attempts"]]
    ta, tb = a["trainer"]._double_q_learning_agent, b["trainer"]._double_q_learning_agent
    assert np.array_equal(ta.Q_table_a, tb.Q_table_a) and np.array_equal(ta.state_action_counter, tb.state_action_counter)
    plain = Trainer(save_path=tmp_path / "plain", seed=5, **kw)
    plain.curriculum_training()
    first = np.load(tmp_path / "a0" / "Q_table_a.npy")
    assert np.array_equal(first, np.load(tmp_path / "plain" / "Q_table_a.npy"))
    assert not np.array_equal(first, np.load(tmp_path / "a1" / "Q_table_a.npy"))
