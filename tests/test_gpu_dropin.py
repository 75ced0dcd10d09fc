"""The drop-in classes (reference names / signatures) on the GPU against the reference's golden vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_training_mdp_replays_reference_trace(golden_dir):
    from dql_multirotor_landing_amd.mdp import CheckResult, ContinuousObservation, Observation, TrainingMdp
    codes = list(CheckResult)
    for level in (0, 2, 4):
        t = np.load(golden_dir / "g2_traces.npz")[f"trace_{level}"]
        m = TrainingMdp(level, 22.92, 20, 4.5)
        n = 0
        for row in t[:500]:
            obs = ContinuousObservation(Observation(rel_p_x=row[2], rel_p_y=row[3], rel_v_x=row[4], rel_a_x=row[5], contact=bool(row[8])),
                                        pitch=row[6], abs_p_z=row[7])
            if int(row[0]) == 0:
                m.reset()
                assert m.discrete_state(obs) == tuple(int(x) for x in row[9:14])
                continue
            a = m.continuous_action(int(row[1]))
            assert a.pitch == row[17] and a.v_z == -0.1
            assert m.discrete_state(obs) == tuple(int(x) for x in row[9:14])
            info = m.check()
            r = m.reward()
            assert codes.index(m._check_result) == int(row[14])
            assert r == row[15]
            assert ("Termination condition" in info) == bool(row[16])
            assert m._cumulative_reward == row[18] and m._step_count == int(row[19]) and m._curriculum_check == int(row[20])
            if "Termination condition" in info:
                assert info["Termination condition"] == m._check_result.value and info["Number of steps"] == m._step_count
            n += 1
        assert n > 300


def test_simulation_mdp_replays_reference_trace(golden_dir):
    from dql_multirotor_landing_amd.mdp import CheckResult, ContinuousObservation, Observation, SimulationMdp
    codes = list(CheckResult)
    t = np.load(golden_dir / "g2s_simulation.npz")["trace"]
    m = SimulationMdp(4, 22.92, 20)
    first = True
    for row in t:
        obs = ContinuousObservation(Observation(rel_p_x=row[3], rel_p_y=row[4], rel_v_x=row[5], rel_v_y=row[6], rel_a_x=row[7], rel_a_y=row[8],
                                                contact=bool(row[12])), pitch=row[9], roll=row[10], abs_p_z=row[11])
        if int(row[0]) == 0:
            m.reset()
        else:
            a = m.continuous_action(int(row[1]), int(row[2]))
            assert a.pitch == row[25] and a.roll == row[26] == 0.0
        sx, sy = m.discrete_state(obs)
        assert sx == tuple(int(x) for x in row[13:18]) and sy == tuple(int(x) for x in row[18:23])
        if int(row[0]) == 1:
            info = m.check()
            assert codes.index(m._check_result) == int(row[23])
            assert ("Termination condition" in info) == bool(row[24])


def test_agent_guess_predict_update_match_reference_stream(golden_dir):
    from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
    g = np.load(golden_dir / "g4_agent.npz")
    agent = DoubleQLearningAgent.load(golden_dir / "assets")
    for eps in (0.0, 0.5, 1.0):
        np.random.seed(42)
        acts = [agent.guess(tuple(int(x) for x in s), eps) for s in g["guess_states"][:200]]
        np.testing.assert_array_equal(acts, g[f"guess_actions_eps{eps}"][:200])
    np.random.seed(42)
    for s in g["guess_states"]:
        agent.get_action(tuple(int(x) for x in s), 0.5)
    assert np.random.uniform(0, 1) == g["guess_rng_next_eps0.5"][0], "MT19937 stream position after 600 guesses (B4)"
    # sequential updates incl. the ignored uniform draw (B1)
    np.random.seed(42)
    a = DoubleQLearningAgent(5)
    n = 300
    for i in range(n):
        a.update(tuple(int(x) for x in g["upd_sa"][i]), tuple(int(x) for x in g["upd_ns"][i]), g["upd_alpha"][i], 0.99, g["upd_reward"][i])
        assert a.Q_table_a[tuple(g["upd_sa"][i])] == g["upd_q_after"][i]
    assert not a.Q_table_b.any() and a.state_action_counter.sum() == n
    # transfer incl. the k = 0 wrap (B6)
    t = DoubleQLearningAgent(5)
    t.Q_table_a = g["tl_Qa_in"].copy(); t.Q_table_b = g["tl_Qb_in"].copy()
    ratios = [1.0, 0.8172650252856599, 0.8211253690681617, 0.8257273369742982, 0.8311571820651724]
    for k in range(5):
        t.transfer_learning(k, ratios[k])
        np.testing.assert_array_equal(t.Q_table_a, g[f"tl_Qa_after{k}"])
        np.testing.assert_array_equal(t.Q_table_b, g[f"tl_Qb_after{k}"])


def test_training_env_single_env_api():
    from dql_multirotor_landing_amd.landing_simulation_env import TrainingLandingEnv, VecLandingEnv
    env = TrainingLandingEnv(0, z_init=4.0, seed=5)
    s = env.reset()
    assert isinstance(s, tuple) and len(s) == 5 and s[0] == 0
    done, steps, total = False, 0, 0.0
    while not done and steps < 500:
        s, r, done, info = env.step(steps % 3)
        assert isinstance(r, float) and isinstance(done, bool) and "Current reward" in info
        steps += 1; total += r
    assert done and "Termination condition" in info and info["Number of steps"] == steps
    with pytest.raises(ValueError):
        env.step(0, 1)
    env.close()
    vec = VecLandingEnv(64, z_init=4.0, seed=5)
    st = vec.reset()
    assert st.shape == (64, 5)
    st, rew, dones, info = vec.step(np.zeros(64, dtype=np.uint8))
    assert rew.shape == (64,) and dones.dtype == bool and not info["was_reset"].any()
    vec.close()


def test_trainer_short_curriculum_run(tmp_path):
    from dql_multirotor_landing_amd.trainer import Trainer
    tr = Trainer(n_envs=1024, save_path=tmp_path / "run", chunk_steps=32, max_steps_per_level=256, checkpoint_every=2)
    hist = tr.curriculum_training()
    assert hist and hist[0]["level"] == 0 and hist[0]["agent_periods"] >= 256 or hist[0]["promoted"]
    assert (tmp_path / "run" / "Q_table_a.npy").exists() and (tmp_path / "Q_table_a.npy").exists()
    assert (tmp_path / "run" / "logs" / "scalars.csv").read_text().count("\n") > 2
    assert tr._double_q_learning_agent.state_action_counter.sum() > 1000


def test_greedy_evaluation_of_reference_tables(golden_dir):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))
    import simulation
    h = simulation.evaluate(golden_dir / "assets", n_envs=512, level=4, max_steps=520)
    assert sum(h.values()) == 512 and h["unfinished"] == 0


def test_error_behaviour_matches_reference_exception_types():
    """SURVEY.md section 8b error convention: the same exception types for the same conditions."""
    from dql_multirotor_landing_amd import ops
    from dql_multirotor_landing_amd.config import DqlConfig, F64
    from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
    from dql_multirotor_landing_amd.engine import Engine
    from dql_multirotor_landing_amd.mdp import ContinuousObservation, Observation, TrainingMdp
    cfg = DqlConfig(dtype=F64)
    # the reference raises ValueError("Unexpected discretization case") for values it cannot bin (NaN here) (pkg/mdp.py:170)
    assert ops.discretise(cfg, [np.nan], [0.0], [0.0], [0.0])[0] == -1
    m = TrainingMdp(0, 22.92, 20, 4.5)
    m.reset()
    with pytest.raises(ValueError):
        m.discrete_state(ContinuousObservation(Observation(rel_p_x=float("nan"))))
    # empty / ragged inputs
    assert len(ops.discretise(cfg, [], [], [], [])) == 0
    with pytest.raises(ValueError):
        ops.discretise(cfg, [0.0, 1.0], [0.0], [0.0], [0.0])
    with pytest.raises(ValueError):
        ops.agent_predict(np.zeros(2835), np.zeros(2835), [945])  # state index out of range
    with pytest.raises(ValueError):
        Engine(DqlConfig(working_curriculum_step=5), 8)
    with pytest.raises(ValueError):
        Engine(DqlConfig(pid_vz=[5.0, 10.0, 1.0, 0.0, 10.0, 10.0]), 8)  # Kd != 0 is outside the fused kernel
    with pytest.raises(ValueError):
        Engine(cfg, 0)
    e = Engine(cfg, 8)
    with pytest.raises(ValueError):
        e.set_tables(np.zeros(10))
    with pytest.raises(ValueError):
        e.step(np.zeros(7, dtype=np.uint8))  # ragged action vector
    reals, ints = e.get_fields()
    assert (ints[0] == -1).all()             # no state yet: the kernel must not follow this index into the tables
    e.train_steps(3, 1.0)                     # ... and does not (reset period first)
    bad = ints.copy(); bad[0, 3] = 945
    with pytest.raises(ValueError):
        e.set_fields(reals, bad)              # state indices address device tables: range-checked at the ABI
    bad[0, 3] = -2
    with pytest.raises(ValueError):
        e.set_fields(reals, bad)
    with pytest.raises(ValueError):
        e.set_curriculum(5)
    with pytest.raises(ValueError):
        e.apply_accum()  # needs windowed accumulation
    e.close()
    a = DoubleQLearningAgent(5)
    with pytest.raises(IndexError):
        a.predict((0, 3, 0, 0, 0))
    with pytest.raises(IndexError):
        a.update((0, 0, 0, 0, 0, 3), (0, 0, 0, 0, 0), 0.1, 0.99, 1.0)


def test_trainer_full_curriculum_hip_equals_oracle_engine(tmp_path, monkeypatch):
    """The whole curriculum loop (ordered promotion rule, transfers, level switches, windowed table exchange) driven by the HIP
    engine and by the CPU oracle behind the same Trainer: same promotions, same episode counts, identical tables."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import dql_multirotor_landing_amd.trainer as T
    from test_dist_gloo import _oracle_engine_class, _strip

    class LocalReducer:
        def __init__(self, eng): self.eng = eng
        def all_reduce(self): self.eng.flush()

    kw = dict(curriculum_steps=5, n_envs=320, chunk_steps=16, sync_period=4, checkpoint_every=10**9, max_num_episodes=400, t_max=4,
              successive_successful_episodes=20, success_rate=0.2, mode="paper", reducer_factory=LocalReducer, judge_envs=320)
    hip = T.Trainer(save_path=tmp_path / "hip", **kw)
    h_hip = _strip(hip.curriculum_training())
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())
    orc = T.Trainer(save_path=tmp_path / "orc", **kw)
    h_orc = _strip(orc.curriculum_training())
    assert h_hip == h_orc and [h["level"] for h in h_hip] == [0, 1, 2, 3, 4]
    assert any(h["promoted"] for h in h_hip) and any(h["exhausted"] for h in h_hip)
    for a, b in zip(hip._engine.get_tables(), orc._engine.get_tables()):
        assert np.array_equal(np.asarray(a).ravel(), np.asarray(b).ravel())
    assert np.array_equal(np.load(tmp_path / "hip" / "Q_table_a.npy"), np.load(tmp_path / "orc" / "Q_table_a.npy"))


def test_trainer_restarts_and_steps_back_alike_on_hip_and_oracle_engines(tmp_path, monkeypatch):
    """Round 5: level restarts (`restart_after`) and steps back (`step_back_after`) — table surgery between launches (counters cleared, slices transferred again, levels
    entered a second time) — driven by the HIP engine and by the CPU oracle behind the same Trainer: same history (restart and step-back counts included), identical tables."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import dql_multirotor_landing_amd.trainer as T
    from test_dist_gloo import _oracle_engine_class, _strip
    kw = dict(curriculum_steps=3, n_envs=192, chunk_steps=16, checkpoint_every=10**9, max_num_episodes=1500, t_max=3, mode="paper", judge_envs=64,
              successive_successful_episodes=20, success_rate=2.0, eps_floor=0.2, restart_after=1.5, step_back_after=1, max_step_backs=2, periods_per_launch=4)
    hip = T.Trainer(save_path=tmp_path / "hip", **kw)
    h_hip = _strip(hip.curriculum_training())
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())
    orc = T.Trainer(save_path=tmp_path / "orc", **kw)
    h_orc = _strip(orc.curriculum_training())
    assert h_hip == h_orc and [h["level"] for h in h_hip] == [0, 1, 2]
    assert h_hip[2]["step_backs"] == 2 and h_hip[2]["restarts"] >= 1 and h_hip[1]["restarts"] >= 1
    for a, b in zip(hip._engine.get_tables(), orc._engine.get_tables()):
        assert np.array_equal(np.asarray(a).ravel(), np.asarray(b).ravel())


def test_sharded_trainer_two_ranks_on_one_gpu_equals_single_process(tmp_path):
    """The sharded Trainer under torch.distributed.run with 2 ranks (tests/_rehearsal_training.py: gloo stand-in communicator, both
    ranks on GPU 0 — the RCCL path needs one GPU per rank and is exercised by the driver's multi-GPU runs and, on one rank, by
    test_gpu_parity.py::test_rccl_reducer_world_size_1): sharded HIP engines + all-reduced counters + gathered
    episode logs give the history and tables of one process running all envs on the same exchange schedule.  Exact for
    sync_period <= 2: launch j acts on tables folded up to launch j-2, so with an exchange every second period no rank ever
    acts on updates the others have not seen; longer windows trade that for fewer exchanges (bounded staleness)."""
    import json, os, socket, subprocess, sys
    from pathlib import Path
    import dql_multirotor_landing_amd.trainer as T
    root = Path(__file__).resolve().parent.parent
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    args = ["--envs", "600", "--mode", "paper", "--chunk", "16", "--sync-period", "2", "--max-episodes", "700", "--levels", "3", "--t-max", "4",
            "--window", "20", "--success-rate", "0.2", "--judge-envs", "400"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(root / "tests" / "_rehearsal_training.py"), "--out", str(tmp_path / "two"), *args],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout[r.stdout.index("{"):])
    assert out["world"] == 2
    strip = lambda hist: [{k: v for k, v in h.items() if not k.startswith("wall")} for h in hist]

    one = T.Trainer(n_envs=600, mode="paper", chunk_steps=16, sync_period=2, max_num_episodes=700, curriculum_steps=3, t_max=4,
                    successive_successful_episodes=20, success_rate=0.2, judge_envs=400, save_path=tmp_path / "one", checkpoint_every=50)
    h1 = json.loads(json.dumps(strip(one.curriculum_training())))
    assert strip(out["history"]) == h1 and [h["level"] for h in h1] == [0, 1, 2]
    for f in ("Q_table_a.npy", "Q_table_b.npy", "state_action_count.npy"):
        assert np.array_equal(np.load(tmp_path / "two" / f), np.load(tmp_path / "one" / f))


def test_attempts_with_two_ranks_on_one_gpu_choose_what_one_process_chooses(tmp_path):
    """Several whole curricula (attempts.py) under torch.distributed.run with 2 ranks: rank 0 alone flies the selection batch, both ranks take the same decision
    (asserted inside the helper), and the attempts — seeds, histories, selection scores, chosen index, chosen tables — are those of one process running all envs."""
    import json, os, socket, subprocess, sys
    from pathlib import Path
    import dql_multirotor_landing_amd.trainer as T
    from dql_multirotor_landing_amd.attempts import SELECTION_SEED, attempt_seed, curriculum_attempts
    from dql_multirotor_landing_amd.evaluation import landing_score
    root = Path(__file__).resolve().parent.parent
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    args = ["--envs", "600", "--mode", "paper", "--chunk", "16", "--sync-period", "2", "--max-episodes", "700", "--levels", "3", "--t-max", "4",
            "--window", "20", "--success-rate", "0.2", "--judge-envs", "400", "--attempts", "3", "--accept-touchdown", "0.0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def two_ranks(tag, accept):
        a = list(args); a[-1] = str(accept)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), str(root / "tests" / "_rehearsal_training.py"), "--out", str(tmp_path / tag), *a],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout[r.stdout.index("{"):])

    def one_process(tag, accept):
        def make(j):
            return T.Trainer(n_envs=600, mode="paper", chunk_steps=16, sync_period=2, max_num_episodes=700, curriculum_steps=3, t_max=4, successive_successful_episodes=20,
                             success_rate=0.2, judge_envs=400, save_path=tmp_path / tag / f"attempt{j}", checkpoint_every=50, seed=attempt_seed(42, j))
        return curriculum_attempts(make, lambda t: landing_score(t._double_q_learning_agent._padded(), 256, 2, seed=SELECTION_SEED, device=0), max_attempts=3, accept_touchdown=accept)

    strip = lambda hist: [{k: v for k, v in h.items() if not k.startswith("wall")} for h in hist]
    # nothing acceptable: all three attempts are trained, the best one is kept
    out, one = two_ranks("two", 2.0), one_process("one", 2.0)
    assert out["world"] == 2 and len(out["attempts"]) == 3 and not out["accepted"] and out["chosen_attempt"] == one["chosen"]
    assert [x["selection"] for x in out["attempts"]] == [x["selection"] for x in one["attempts"]]
    assert len({x["selection"]["touchdown_rate"] for x in out["attempts"]} | {x["selection"]["goal_hold_rate"] for x in out["attempts"]}) > 2  # (the attempts differ)
    assert strip(out["history"]) == json.loads(json.dumps(strip(one["history"])))
    for j in range(3):
        for f in ("Q_table_a.npy", "Q_table_b.npy", "state_action_count.npy"):
            assert np.array_equal(np.load(tmp_path / "two" / f"attempt{j}" / f), np.load(tmp_path / "one" / f"attempt{j}" / f))
    # anything acceptable: the first attempt is taken and the others are never trained
    out0 = two_ranks("two0", 0.0)
    assert out0["accepted"] and out0["chosen_attempt"] == 0 and len(out0["attempts"]) == 1 and not (tmp_path / "two0" / "attempt1").exists()


def test_g13_reference_env_outputs_through_the_hip_engine(golden_dir):
    """a17-a19 on the GPU: the fixture's fake Gazebo played back a flight of this simulator to the REFERENCE's TrainingLandingEnv /
    SimulationLandingEnv; here the HIP engine (one env, float64, same seed and actions) flies it again: identical signals, and per
    agent period the reference env's returned state / reward / done / CheckResult == the fused step's."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_oracle_golden import G13_CASES, _g13_check, _g13_fly
    from dql_multirotor_landing_amd.engine import Engine
    z = np.load(golden_dir / "g13_env.npz")
    for tag in G13_CASES:
        sig, res = _g13_fly(Engine, tag, z)
        _g13_check(tag, z, sig, res)


def test_g13_dropin_env_classes_return_what_the_reference_env_returned(golden_dir):
    """The drop-in TrainingLandingEnv / SimulationLandingEnv (same constructor arguments, same reset() / step() protocol) on the
    fixture's seed and actions return, call by call, what the reference's classes returned."""
    from dql_multirotor_landing_amd.landing_simulation_env import SimulationLandingEnv, TrainingLandingEnv
    z = np.load(golden_dir / "g13_env.npz")
    for tag, cls, level, t_max, seed in (("train0", TrainingLandingEnv, 0, 4, 1300), ("train2", TrainingLandingEnv, 2, 4, 1302), ("sim4", SimulationLandingEnv, 4, 6, 1304)):
        rows, actions = z[f"{tag}_rows"], z[f"{tag}_actions"]
        env = cls(level, t_max=t_max, z_init=4.0, seed=seed)
        i = 0
        while i < len(rows):
            assert rows[i, 0] == 0
            s = env.reset()
            if cls is TrainingLandingEnv:
                assert tuple(s) == tuple(int(v) for v in rows[i, 2:7])
            else:
                assert tuple(s[0]) == tuple(int(v) for v in rows[i, 2:7]) and tuple(s[1]) == tuple(int(v) for v in rows[i, 7:12])
            i += 1
            while i < len(rows) and rows[i, 0] == 1:
                if cls is TrainingLandingEnv:
                    st, r, done, info = env.step(int(actions[i]))
                    assert tuple(st) == tuple(int(v) for v in rows[i, 2:7]) and r == rows[i, 12] and done == bool(rows[i, 13])
                    assert info["Current reward"] == r and (("Termination condition" in info) == done)
                else:
                    sx, sy, done, info = env.step(int(actions[i]), 2)
                    assert tuple(sx) == tuple(int(v) for v in rows[i, 2:7]) and tuple(sy) == tuple(int(v) for v in rows[i, 7:12]) and done == bool(rows[i, 13])
                i += 1
                if done:
                    break
        env.close()


def test_trainer_resume_on_the_hip_engine_equals_uninterrupted_run(tmp_path):
    """Trainer.save -> Trainer.load -> curriculum_training() on the REAL engine (8 periods per launch, windowed table schedule,
    env state + period index + promotion bookkeeping from the tagged checkpoint files): history and tables equal the uninterrupted run."""
    import json
    import dql_multirotor_landing_amd.trainer as T
    kw = dict(curriculum_steps=3, n_envs=2048, chunk_steps=16, checkpoint_every=3, max_num_episodes=6000, t_max=3, periods_per_launch=8, sync_period=8,
              successive_successful_episodes=10, success_rate=0.25, mode="paper", judge_envs=48, eps_floor=0.3)
    strip = lambda hist: [{k: v for k, v in h.items() if not k.startswith("wall")} for h in hist]
    full = T.Trainer(save_path=tmp_path / "full" / "01-01-2026 10:00:00", **kw)
    h_full = strip(full.curriculum_training())
    assert len(h_full) == 3

    class Stop(Exception):
        pass

    part = T.Trainer(save_path=tmp_path / "part" / "01-01-2026 10:00:00", **kw)
    n_saves = {"n": 0}
    real_save = part.save
    def save_then_stop():
        real_save()
        if part._progress is not None and part._progress["level"] == 1:
            n_saves["n"] += 1
            if n_saves["n"] == 1:
                raise Stop()
    part.save = save_then_stop
    with pytest.raises(Stop):
        part.curriculum_training()
    part._engine.close()
    st = json.loads((tmp_path / "part" / "01-01-2026 10:00:00" / "trainer.json").read_text())
    assert st["progress"]["level"] == 1 and st["progress"]["tag"]["world"] == 1
    back = T.Trainer.load(tmp_path / "part")
    h_back = strip(back.curriculum_training())
    assert json.loads(json.dumps(h_back)) == json.loads(json.dumps(h_full))
    for a, b in zip(back._engine.get_tables(), full._engine.get_tables()):
        np.testing.assert_array_equal(a, b)
    full._engine.close(); back._engine.close()


def test_agent_tables_are_live_host_arrays():
    """`Q_table_a / Q_table_b / state_action_counter` stay ordinary public arrays while the arithmetic runs on tables resident on the
    device (include/dql.h dql_agent_mirror_*): cells written in place, attributes rebound to other arrays (any dtype / layout) and
    tables of fewer than 5 levels are all honoured by the next call; every update lands in the host arrays; predict answers on the
    tables as they are now — against the CPU oracle replaying the same sequence, both update rules."""
    from dql_multirotor_landing_amd.config import Q_PAPER, Q_REFERENCE
    from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
    from oracle import oracle as orc
    rng = np.random.default_rng(11)
    for mode, quirks in (("reference", Q_REFERENCE), ("paper", Q_PAPER)):
        for levels in (5, 3, 1):
            np.random.seed(levels)
            a = DoubleQLearningAgent(levels, mode=mode)
            shape = (levels, 3, 3, 3, 7, 3)
            a.Q_table_a = rng.normal(size=shape)
            a.Q_table_b = rng.normal(size=shape).astype(np.float32)              # not float64: replaced by a float64 copy on first use
            a.state_action_counter = np.asfortranarray(rng.integers(0, 9, shape).astype(np.float64))  # not C order: likewise
            o = [np.zeros(2835), np.zeros(2835), np.zeros(2835)]
            for k, t in enumerate((a.Q_table_a, a.Q_table_b, a.state_action_counter)):
                o[k][:levels * 567] = np.asarray(t, dtype=np.float64).ravel()
            ns = None
            for i in range(150):
                sa = tuple(int(rng.integers(0, d)) for d in shape)
                if ns is not None and i % 3:
                    sa = ns + (sa[5],)                                               # the loop's shape: the next transition starts where the last ended
                ns = tuple(int(rng.integers(0, d)) for d in shape[:5])
                if i % 17 == 5:
                    a.Q_table_a[ns + (1,)] = 3.25; o[0][np.ravel_multi_index(ns + (1,), shape)] = 3.25      # written in place between calls
                if i % 29 == 7:
                    a.Q_table_b = a.Q_table_b * 0.5; o[1] *= 0.5                  # rebound to a new array
                if i % 41 == 9:
                    a.state_action_counter[...] = 0.0; o[2][:] = 0.0
                alpha, reward, done = float(rng.uniform(0.02, 1.0)), float(rng.normal() * 5), bool(rng.integers(0, 2))
                st = np.random.get_state(); coin = not np.random.uniform(0, 1) < 0.5; np.random.set_state(st)  # the draw update() is about to make
                a.update(sa, ns, alpha, 0.99, reward, done=done)
                orc.agent_update(o[0], o[1], o[2], np.array([np.ravel_multi_index(sa, shape)], np.int32), np.array([np.ravel_multi_index(ns, shape[:5])], np.int32),
                                 np.array([alpha]), 0.99, np.array([reward]), quirks=quirks, coin=np.array([coin], np.uint8), done=np.array([done], np.uint8))
                for got, want in zip((a.Q_table_a, a.Q_table_b, a.state_action_counter), o):
                    assert got.dtype == np.float64 and got.shape == shape
                    np.testing.assert_array_equal(got.ravel(), want[:levels * 567])
                assert not any(w[levels * 567:].any() for w in o)
                q = ns if i % 2 else tuple(int(rng.integers(0, d)) for d in shape[:5])
                assert a.predict(q) == orc.agent_predict(o[0], o[1], np.array([np.ravel_multi_index(q, shape[:5])], np.int32))[0]
            with pytest.raises(IndexError):
                a.predict((levels, 0, 0, 0, 0))
            a.Q_table_a = np.zeros((levels + 1, 3, 3, 3, 7, 3))
            with pytest.raises(ValueError):
                a.predict((0, 0, 0, 0, 0))
            a.close()


def test_training_env_float32_keyword():
    """`TrainingLandingEnv(dtype=F32)` (build-specific keyword): the same API on the float32 step.  Same seed, same scripted actions: the first
    periods of an episode give the same discrete states as the float64 env except where an observation sits within rounding of a bin edge."""
    from dql_multirotor_landing_amd.config import F32
    from dql_multirotor_landing_amd.landing_simulation_env import TrainingLandingEnv
    e64 = TrainingLandingEnv(0, z_init=4.0, seed=11)
    e32 = TrainingLandingEnv(0, z_init=4.0, seed=11, dtype=F32)
    rng = np.random.default_rng(3)
    same = total = 0
    for _ in range(6):
        s64, s32 = e64.reset(), e32.reset()
        assert len(s32) == 5 and all(isinstance(x, int) for x in s32)
        same += s64 == s32; total += 1
        for _ in range(25):
            a = int(rng.integers(0, 3))
            o64, o32 = e64.step(a), e32.step(a)
            assert np.isfinite(o32[1]) and isinstance(o32[2], bool) and isinstance(o32[3], dict)
            same += o64[0] == o32[0]; total += 1
            if o64[2] or o32[2]:
                break
    assert same / total > 0.9, (same, total)
    with pytest.raises(ValueError):
        e32.step(0, 1)
    e64.close(); e32.close()


def test_agent_update_is_deferred_and_every_access_sees_it_finished(golden_dir):
    """ABI v6: `DoubleQLearningAgent.update` returns when its kernel is launched; the changed cell and its visit counter reach the public arrays at the next access
    through the agent.  Whatever that access is — reading a table, replacing one, predict, another update, save, close — it sees the update done, and the
    sequence of table values equals the reference's (golden G4: 300 sequential updates incl. the ignored uniform draw B1), checked after every single update."""
    from dql_multirotor_landing_amd.double_q_learning import DoubleQLearningAgent
    g = np.load(golden_dir / "g4_agent.npz")
    np.random.seed(42)
    a = DoubleQLearningAgent(5)
    for i in range(120):
        sa = tuple(int(x) for x in g["upd_sa"][i])
        a.update(sa, tuple(int(x) for x in g["upd_ns"][i]), g["upd_alpha"][i], 0.99, g["upd_reward"][i])
        assert a._pending                                    # nothing has asked for the result yet
        kind = i % 4
        if kind == 0:
            assert a.Q_table_a[sa] == g["upd_q_after"][i]    # a read through the agent
        elif kind == 1:
            a.predict(sa[:5]); assert a._qa[sa] == g["upd_q_after"][i]   # any call on the agent (the raw array is patched by then)
        elif kind == 2:
            held = a._qa; a.Q_table_b = np.zeros_like(a._qb); assert held[sa] == g["upd_q_after"][i]   # replacing a table finishes the update first
        # kind 3: straight into the next update, which finishes this one before it launches
        assert not a._pending or kind == 3
    assert a.state_action_counter.sum() == 120 and not a.Q_table_b.any()
    # two updates back to back, then close(): the arrays hold both
    b = DoubleQLearningAgent(5)
    b.update((0, 1, 1, 1, 3, 2), (0, 1, 1, 1, 3), 0.5, 0.99, 1.0)
    b.update((0, 1, 1, 1, 3, 1), (0, 1, 1, 1, 3), 0.5, 0.99, 2.0)
    qa = b._qa
    b.close()
    assert qa[0, 1, 1, 1, 3, 2] == 0.5 and qa[0, 1, 1, 1, 3, 1] == 1.0 + 0.5 * 0.99 * 0.0 and b._cnt.sum() == 2
