"""world_size-2 coverage of the multi-GPU path on CPU (gloo): env sharding by global env id, the int64 window
all-reduce and the base/work table semantics of dql_multirotor_landing_amd.dist.ShardedRunner.  The compute engine
injected here is the CPU oracle and the communicator a torch/gloo stand-in (tests/_torch_comm.py; both allowed in tests
only); on the GPU the same runner drives the HIP Engine with the RCCL reducer of libdql_hip.so."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
N_TOTAL, STEPS = 96, 40


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, sync_period, out_dir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    from _torch_comm import HostWindowReducer
    from dql_multirotor_landing_amd.config import DqlConfig, F64
    from dql_multirotor_landing_amd.dist import ShardedRunner, shard_range
    from oracle.oracle import Oracle
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_range(N_TOTAL, rank, world)
    eng = Oracle(DqlConfig(dtype=F64), hi - lo, seed=42, env_id_offset=lo)
    run = ShardedRunner(eng, HostWindowReducer(eng), sync_period=sync_period)
    run.train_steps(STEPS // 2, 1.0)
    run.train_steps(STEPS - STEPS // 2, 0.2)
    run.sync()
    np.savez(Path(out_dir) / f"rank{rank}.npz", qa=eng.qa, count=eng.count, ints=eng.get_fields()[1], reals=eng.get_fields()[0])
    dist.barrier()
    dist.destroy_process_group()


def _run_world(tmp_path, sync_period):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, sync_period, str(tmp_path)), nprocs=2, join=True)
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]


def test_shard_range_covers_everything():
    from dql_multirotor_landing_amd.dist import shard_range
    for n, w in ((96, 2), (97, 4), (5, 8), (262144, 8)):
        r = [shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def test_two_ranks_sync_every_step_equals_single_process(tmp_path):
    """Rank-count invariance at K = 1: 2 ranks x 48 envs == 1 process x 96 envs on the same sync schedule, bit for bit
    (integer sums are order independent and the RNG is keyed by global env id)."""
    from dql_multirotor_landing_amd.config import DqlConfig, F64
    from dql_multirotor_landing_amd.dist import ShardedRunner
    from oracle.oracle import Oracle
    ranks = _run_world(tmp_path, sync_period=1)

    class LocalReducer:  # world size 1: the sum over ranks is the identity
        def all_reduce(self):
            single.flush()

    single = Oracle(DqlConfig(dtype=F64), N_TOTAL, seed=42)
    run = ShardedRunner(single, LocalReducer(), sync_period=1)
    run.train_steps(STEPS // 2, 1.0); run.train_steps(STEPS - STEPS // 2, 0.2); run.sync()
    for r in ranks:
        np.testing.assert_array_equal(r["qa"], single.qa)
        np.testing.assert_array_equal(r["count"], single.count)
    reals, ints = single.get_fields()
    np.testing.assert_array_equal(np.concatenate([ranks[0]["ints"], ranks[1]["ints"]], axis=1), ints)
    np.testing.assert_array_equal(np.concatenate([ranks[0]["reals"], ranks[1]["reals"]], axis=1), reals)
    assert single.count.sum() > 0


def test_two_ranks_windowed_equals_in_process_emulation(tmp_path):
    """K = 8: ranks act on their local work tables inside a window; at the sync every rank holds identical tables,
    equal to an in-process emulation of the same semantics."""
    from dql_multirotor_landing_amd.config import DqlConfig, F64
    from dql_multirotor_landing_amd.dist import shard_range
    from oracle.oracle import Oracle
    ranks = _run_world(tmp_path, sync_period=8)
    np.testing.assert_array_equal(ranks[0]["qa"], ranks[1]["qa"])
    np.testing.assert_array_equal(ranks[0]["count"], ranks[1]["count"])
    shards = []
    for k in range(2):
        lo, hi = shard_range(N_TOTAL, k, 2)
        o = Oracle(DqlConfig(dtype=F64), hi - lo, seed=42, env_id_offset=lo); o.set_windowed(True); shards.append(o)
    done = 0
    for eps, n in ((1.0, STEPS // 2), (0.2, STEPS - STEPS // 2)):
        for _ in range(n):
            for o in shards:
                o.train_steps(1, eps)
            done += 1
            if done % 8 == 0:
                tot = shards[0].get_accum() + shards[1].get_accum()
                for o in shards:
                    o.set_accum(tot); o.apply_accum()
    if done % 8:
        tot = shards[0].get_accum() + shards[1].get_accum()
        for o in shards:
            o.set_accum(tot); o.apply_accum()
    np.testing.assert_array_equal(ranks[0]["qa"], shards[0].qa)
    np.testing.assert_array_equal(ranks[1]["count"], shards[1].count)


# ---- the whole curriculum loop (BASELINE config 4 in miniature): Trainer on 2 ranks == Trainer on 1 process ----
TR_KW = dict(curriculum_steps=3, n_envs=96, chunk_steps=8, sync_period=2, checkpoint_every=10**9, max_num_episodes=150, t_max=3,
             successive_successful_episodes=10, success_rate=0.25, mode="paper", judge_envs=70)


def _oracle_engine_class():
    from dql_multirotor_landing_amd.config import CHECK_NAMES
    from oracle.oracle import Oracle

    class OracleEngine:  # the Engine surface the Trainer uses, computed by the CPU oracle (tests only)
        def __init__(self, cfg, n, seed=42, device=0, env_id_offset=0):
            self.o = Oracle(cfg, n, seed=seed, env_id_offset=env_id_offset)
            self.n = n
        def __getattr__(self, name):
            return getattr(self.o, name)
        def step_index(self):
            return self.o.step_index
        def set_step_index(self, j):
            self.o.publish_tables(); self.o.step_index = int(j)
        def get_tables(self):
            return self.o.qa.copy(), self.o.qb.copy(), self.o.count.copy()
        def stats(self):
            d = self.o.stats_dict()
            d["by_code"] = {CHECK_NAMES[i]: d["by_code"][i] for i in range(len(CHECK_NAMES))}
            return d
    return OracleEngine


def _strip(hist):
    return [{k: v for k, v in h.items() if not k.startswith("wall")} for h in hist]


def _trainer_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import json
    import torch.distributed as dist
    from _torch_comm import TorchComm
    import dql_multirotor_landing_amd.trainer as T
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    T.Engine = _oracle_engine_class()
    tr = T.Trainer(save_path=Path(out_dir) / "run", comm=TorchComm(), **TR_KW)
    assert tr._world == 2 and tr._rank == rank
    hist = tr.curriculum_training()
    qa, qb, cnt = tr._engine.get_tables()
    np.savez(Path(out_dir) / f"trainer_rank{rank}.npz", qa=qa, count=cnt)
    (Path(out_dir) / f"hist{rank}.json").write_text(json.dumps(_strip(hist)))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_curriculum_two_ranks_equals_single_process(tmp_path, monkeypatch):
    """Same promotions (episode and agent period), same episode counts and bit-identical tables whether the 96 envs run on one
    process or are sharded over two ranks: counters are summed, episode logs gathered in global env order, tables exchanged
    on the same schedule."""
    import json
    import torch.multiprocessing as mp
    import dql_multirotor_landing_amd.trainer as T
    mp.spawn(_trainer_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    h0, h1 = (json.loads((tmp_path / f"hist{r}.json").read_text()) for r in range(2))
    assert h0 == h1 and [h["level"] for h in h0] == [0, 1, 2]
    assert (tmp_path / "run" / "Q_table_a.npy").exists()  # rank 0 wrote the checkpoint
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())

    # world size 1 with the same sync_period: the Trainer puts itself on the windowed schedule (dist.LocalWindowReducer)
    single = T.Trainer(save_path=tmp_path / "single", **TR_KW)
    hs = json.loads(json.dumps(_strip(single.curriculum_training())))
    assert hs == h0
    assert any(h["promoted"] for h in hs) and sum(h["episodes"] for h in hs) > 0
    qa, _, cnt = single._engine.get_tables()
    for r in range(2):
        z = np.load(tmp_path / f"trainer_rank{r}.npz")
        np.testing.assert_array_equal(z["qa"], qa)
        np.testing.assert_array_equal(z["count"], cnt)


def test_trainer_resume_equals_uninterrupted_run(tmp_path, monkeypatch):
    """Stop after a mid-level checkpoint, Trainer.load, finish: history and tables equal the uninterrupted run (oracle engine).
    The checkpoint carries every env's simulator state, the period index (RNG counters, tick schedule), the level's episode
    count (exploration schedule) and the promotion bookkeeping; the level's transfer is not applied a second time."""
    import json
    import dql_multirotor_landing_amd.trainer as T
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())
    kw = dict(curriculum_steps=3, n_envs=64, chunk_steps=8, checkpoint_every=3, max_num_episodes=120, t_max=3,
              successive_successful_episodes=10, success_rate=0.25, mode="paper", judge_envs=48, eps_floor=0.3)
    for sync in (None, 2):
        full = T.Trainer(save_path=tmp_path / f"full{sync}" / "01-01-2026 10:00:00", sync_period=sync, **kw)
        h_full = _strip(full.curriculum_training())
        assert len(h_full) == 3 and sum(h["agent_periods"] for h in h_full) > 3 * 8 * 3  # several checkpoints were taken

        class Stop(Exception):
            pass

        part = T.Trainer(save_path=tmp_path / f"part{sync}" / "01-01-2026 10:00:00", sync_period=sync, **kw)
        n_saves = {"n": 0}
        real_save = part.save
        def save_then_stop():
            real_save()
            if part._progress is not None and part._progress["level"] == 1:
                n_saves["n"] += 1
                if n_saves["n"] == 1:
                    raise Stop()  # "power cut" right after the first mid-level checkpoint of level 1
        part.save = save_then_stop
        with pytest.raises(Stop):
            part.curriculum_training()
        st = json.loads((tmp_path / f"part{sync}" / "01-01-2026 10:00:00" / "trainer.json").read_text())
        assert st["progress"]["level"] == 1 and st["build"]["eps_floor"] == 0.3 and st["build"]["judge_envs"] == 48 and st["build"]["sync_period"] == sync
        back = T.Trainer.load(tmp_path / f"part{sync}")
        assert back._working_curriculum_step == 1 and back._eps_floor == 0.3 and back._chunk_steps == 8
        h_back = _strip(back.curriculum_training())
        assert json.loads(json.dumps(h_back)) == json.loads(json.dumps(h_full))
        for a, b in zip(back._engine.get_tables(), full._engine.get_tables()):
            np.testing.assert_array_equal(a, b)


def test_checkpoint_files_carry_one_tag_and_a_mixed_checkpoint_is_not_resumed(tmp_path, monkeypatch):
    """every file of a checkpoint is written as temp + replace, the env-state shard and trainer.json carry the same
    (level, agent periods, chunk, world, generation) tag, and a shard that belongs to ANOTHER checkpoint (a rank killed between its
    shard and rank 0's trainer.json, a leftover) is not flown: the level resumes from its tables, with a warning."""
    import json
    import shutil
    import dql_multirotor_landing_amd.trainer as T
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())
    kw = dict(curriculum_steps=2, n_envs=32, chunk_steps=8, checkpoint_every=2, max_num_episodes=60, t_max=3,
              successive_successful_episodes=10, success_rate=0.25, mode="paper", judge_envs=16, eps_floor=0.3)

    class Stop(Exception):
        pass

    run = tmp_path / "a" / "01-01-2026 10:00:00"
    tr = T.Trainer(save_path=run, **kw)
    saves = {"n": 0}
    real_save = tr.save
    def save_hook():
        real_save()
        if tr._progress is not None:
            saves["n"] += 1
            if saves["n"] == 1:
                shutil.copy(run / "env_state_rank0.npz", tmp_path / "older_shard.npz")
            if saves["n"] == 2:
                raise Stop()
    tr.save = save_hook
    with pytest.raises(Stop):
        tr.curriculum_training()
    st = json.loads((run / "trainer.json").read_text())
    z = np.load(run / "env_state_rank0.npz")
    tag = st["progress"]["tag"]
    assert {k: int(z["tag_" + k]) for k in tag} == tag and tag["world"] == 1 and tag["generation"] == 2 and tag["chunk_i"] == st["progress"]["chunk_i"]
    assert not list(run.glob(".*tmp*")) and not list(tmp_path.glob("a/.*tmp*"))  # no temp file left behind
    # the shard of the EARLIER checkpoint beside the newer trainer.json = what a kill between the two writes leaves
    shutil.copy(tmp_path / "older_shard.npz", run / "env_state_rank0.npz")
    back = T.Trainer.load(tmp_path / "a")
    with pytest.warns(RuntimeWarning, match="belongs to checkpoint"):
        hist = back.curriculum_training()
    assert [h["level"] for h in hist][-1] == 1
    # ... and a finished run loads as finished: curriculum_training() returns the stored history instead of asking for level 2 of 2
    done = T.Trainer.load(tmp_path / "a")
    assert done._working_curriculum_step == 2
    assert _strip(done.curriculum_training()) == json.loads(json.dumps(_strip(hist)))


def test_reducer_factory_without_sync_period_defaults_to_two(tmp_path, monkeypatch):
    import dql_multirotor_landing_amd.trainer as T
    from dql_multirotor_landing_amd.dist import LocalWindowReducer
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())
    tr = T.Trainer(save_path=tmp_path / "r", curriculum_steps=1, n_envs=16, chunk_steps=8, max_num_episodes=20, t_max=3, mode="paper",
                   reducer_factory=LocalWindowReducer, checkpoint_every=10**9)
    eng, runner = tr._make_engine(tr._config(0))
    assert runner.sync_period == 2 and runner.reducer is not None


def test_resume_with_a_mismatching_shard_stops_every_rank_after_the_vote(tmp_path, monkeypatch):
    """an env-state shard holding another env count used to raise on ITS rank before the all-reduce of the resume vote, leaving the
    other ranks in the collective.  Now the mismatch is part of the vote: every rank takes part in the collective and every rank raises."""
    import dql_multirotor_landing_amd.trainer as T
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())

    class FakeComm:  # the other rank's votes are added to this rank's, as the all-reduce would
        def __init__(self, other):
            self.other, self.calls, self.rank, self.world = other, 0, 0, 2
        def all_reduce_sum(self, v):
            self.calls += 1
            return np.asarray(v, dtype=np.float64) + np.asarray(self.other, dtype=np.float64)
        def barrier(self):
            pass

    kw = dict(curriculum_steps=1, n_envs=32, chunk_steps=8, checkpoint_every=10**9, max_num_episodes=20, t_max=3, mode="paper")
    tr = T.Trainer(save_path=tmp_path / "r" / "01-01-2026 10:00:00", **kw)
    eng, _ = tr._make_engine(tr._config(0))
    tr._engine = eng
    tr._progress = {"level": 0, "steps": 8, "chunk_i": 1}
    tr.save()
    progress = dict(tr._progress, tag=tr._checkpoint_tag())
    # (a) this rank's shard is fine, the OTHER rank reports a shard of another env count: this rank must not fly its shard either
    tr._comm, tr._world = FakeComm([1.0, 1.0]), 2
    progress["tag"]["world"] = 1  # the shard was written by a world of 1; keep the tags equal so that only the vote decides
    with pytest.raises(ValueError, match="another rank"):
        tr._restore_env_state(eng, progress)
    assert tr._comm.calls == 1
    # (b) the other rank merely has no shard (not an error): nobody flies, nobody raises
    tr._comm = FakeComm([1.0, 0.0])
    assert tr._restore_env_state(eng, progress) is False and tr._comm.calls == 1
    # (c) this rank's own shard holds another env count: it still votes first, then raises with its own message
    small, _ = T.Trainer(save_path=tmp_path / "s" / "01-01-2026 10:00:00", **dict(kw, n_envs=16))._make_engine(tr._config(0))
    tr._comm = FakeComm([0.0, 0.0])
    with pytest.raises(ValueError, match="holds 32 envs"):
        tr._restore_env_state(small, progress)
    assert tr._comm.calls == 1
    # (d) all good on both ranks: the state is restored
    tr._comm = FakeComm([0.0, 0.0])
    assert tr._restore_env_state(eng, progress) is True


# ---- round 5: the first multi-GPU run must answer every open question by itself: both exchanges timed in ONE run,
# a hash of the three tables compared across ranks after the final exchange, and a failing exchange must not strand the other ranks ----
def _exchange_worker(rank, world, port, out_dir, case):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import json
    import torch.distributed as dist
    from _torch_comm import HostWindowReducer, TorchComm
    from dql_multirotor_landing_amd.config import DqlConfig, F64
    from dql_multirotor_landing_amd.dist import compare_exchanges, replicas_identical, shard_range, timed_region
    OracleEngine = _oracle_engine_class()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = TorchComm()
    lo, hi = shard_range(N_TOTAL, rank, world)
    eng = OracleEngine(DqlConfig(dtype=F64), hi - lo, seed=42, env_id_offset=lo)
    eng.sync = lambda: None
    res = {}

    class Failing:   # a peer-to-peer exchange that gives up on a peer.  Like the real one it involves NO host collective (every rank's device-side wait is
        calls = 0    # bounded by itself), so a rank that fails leaves nobody waiting inside the exchange; the failure surfaces on ONE rank only
        def __init__(self, engine):
            self.engine = engine
        def all_reduce(self):
            Failing.calls += 1
            self.engine.flush()
            if rank == 1 and Failing.calls > 2:
                raise RuntimeError("peer-to-peer table exchange gave up waiting for a peer")

    if case == "both":
        res["legs"] = compare_exchanges(eng, comm, {"rccl": HostWindowReducer(eng), "p2p": HostWindowReducer(eng)}, "rccl", sync_period=4, steps=12, warmup=4, eps=0.5, reps=3)
        res["identical_after"] = replicas_identical(eng, comm)
        if rank == 1:   # one rank's replica drifts (what a half-applied exchange would leave behind): the check must see it on EVERY rank
            eng.o._qa[7] += 1e-9
        res["identical_after_drift"] = replicas_identical(eng, comm)
    elif case == "failing":
        res["legs"] = compare_exchanges(eng, comm, {"rccl": HostWindowReducer(eng), "p2p": Failing(eng)}, "rccl", sync_period=4, steps=12, warmup=4, eps=0.5, reps=3,
                                        log=lambda m: None)
        res["after"] = timed_region(eng, comm, HostWindowReducer(eng), 4, 8, 0, 0.5, reps=1) is not None   # the communicator is still usable: nobody hangs
    (Path(out_dir) / f"ex{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["both", "failing"])
def test_both_exchanges_in_one_run_and_replica_check(tmp_path, case):
    import json
    import torch.multiprocessing as mp
    mp.spawn(_exchange_worker, args=(2, _free_port(), str(tmp_path), case), nprocs=2, join=True)
    r0, r1 = (json.loads((tmp_path / f"ex{r}.json").read_text()) for r in range(2))
    if case == "both":
        for r in (r0, r1):
            assert list(r["legs"]) == ["rccl", "p2p"]                                    # the primary exchange first, then the other, same engine
            for leg in r["legs"].values():
                assert leg["replicas_identical"] is True and leg["value"] > 0 and leg["value_min"] <= leg["value"] <= leg["value_max"]
            assert r["identical_after"] is True and r["identical_after_drift"] is False   # a 1e-9 drift in ONE cell of ONE rank is seen by both
        assert r0["legs"]["rccl"]["value"] == r1["legs"]["rccl"]["value"]                 # MAX over ranks / SUM of env-steps: one figure for the job
    else:
        for r in (r0, r1):
            assert r["legs"]["rccl"]["replicas_identical"] is True
            assert "skipped" in r["legs"]["p2p"] and "failed" in r["legs"]["p2p"]["skipped"]   # on BOTH ranks, although only rank 1 raised
            assert r["after"] is True


def test_tables_fingerprint_is_exact_in_float64():
    from dql_multirotor_landing_amd.dist import tables_fingerprint
    a = np.arange(2835.0); b = a.copy(); b[100] = np.nextafter(b[100], 1e9)
    f1, f2 = tables_fingerprint(a, a, a), tables_fingerprint(a, b, a)
    assert f1 != f2 and all(float(int(x)) == x and 0 <= x < 2 ** 32 for x in f1 + f2)
    assert tables_fingerprint(a.reshape(5, 567), a, a) == f1   # bytes, not shapes


def test_level_restart_on_plateau(tmp_path, monkeypatch):
    """Round 5, Trainer(restart_after=...): a level k >= 1 that has not been promoted after that many episodes per env is started over — its slice
    transferred from level k - 1 again, its visit counters cleared, the promotion windows emptied; when the budget runs out the level hands over
    the tables of its BEST attempt.  The promotion rule itself is untouched (an unreachable success rate never promotes); None = the reference."""
    import dql_multirotor_landing_amd.trainer as T
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())
    kw = dict(curriculum_steps=2, n_envs=48, chunk_steps=8, checkpoint_every=10**9, max_num_episodes=400, t_max=3, mode="paper", judge_envs=16,
              successive_successful_episodes=10, eps_floor=0.3)
    a = T.Trainer(save_path=tmp_path / "a", success_rate=2.0, restart_after=1.5, **kw)
    ha = a.curriculum_training()
    assert [h["promoted"] for h in ha] == [False, False] and ha[0]["restarts"] == 0 and ha[1]["restarts"] >= 3   # level 0 has nothing to restart from
    b = T.Trainer(save_path=tmp_path / "b", success_rate=2.0, **kw)
    hb = b.curriculum_training()
    assert [h["restarts"] for h in hb] == [0, 0]
    qa_a, _, cnt_a = a._engine.get_tables(); qa_b, _, cnt_b = b._engine.get_tables()
    assert np.isfinite(qa_a).all() and not np.array_equal(qa_a, qb_ := qa_b)                 # the restarts changed the course of level 1
    np.testing.assert_array_equal(np.asarray(qa_a).reshape(5, -1)[2:], 0.0)                  # and touched no level above it
    assert np.asarray(cnt_a).reshape(5, -1)[1].sum() < np.asarray(cnt_b).reshape(5, -1)[1].sum()   # counters of the level were cleared on the way


def test_step_back_after_failed_restarts(tmp_path, monkeypatch):
    """Round 5, Trainer(step_back_after=...): a level k >= 2 that has used up that many restarts steps BACK — level k - 1 is started over from level k - 2
    and level k after it (the restarts had been redrawing level k from one and the same table of level k - 1).  One history entry per level whatever the
    path, `step_backs` counts them, at most `max_step_backs` per run; level 1 never steps back (level 0 is not restarted); without the keyword nothing changes."""
    import dql_multirotor_landing_amd.trainer as T
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())
    kw = dict(curriculum_steps=3, n_envs=48, chunk_steps=8, checkpoint_every=10**9, max_num_episodes=500, t_max=3, mode="paper", judge_envs=16,
              successive_successful_episodes=10, eps_floor=0.3, success_rate=2.0, restart_after=1.5)
    a = T.Trainer(save_path=tmp_path / "a", step_back_after=1, max_step_backs=2, **kw)
    ha = a.curriculum_training()
    assert [h["level"] for h in ha] == [0, 1, 2] and not any(h["promoted"] for h in ha)
    assert ha[2]["step_backs"] == 2 and ha[1]["step_backs"] >= 1 and ha[0]["step_backs"] == 0   # level 2 gave up twice; level 1's entry is of its last run
    assert ha[2]["exhausted"] and ha[2]["restarts"] >= 2                                        # once the steps back are spent the level restarts to its budget's end
    b = T.Trainer(save_path=tmp_path / "b", **kw)
    hb = b.curriculum_training()
    assert [h["step_backs"] for h in hb] == [0, 0, 0]
    assert np.isfinite(a._engine.get_tables()[0]).all()
    with pytest.raises(ValueError, match="step_back_after"):
        T.Trainer(save_path=tmp_path / "c", step_back_after=1, **{**kw, "restart_after": None})


def test_step_back_falls_back_to_the_promoted_tables_when_the_relearning_fails(tmp_path, monkeypatch):
    """The level learnt again after a step back has a short budget, never steps back itself, and — when the rule does not promote it — the run goes on from the tables
    the level WAS promoted with: its history entry stays the promoted one.  Scripted promotion window: levels 0 and 1 pass on their first attempt, nothing passes after."""
    import dql_multirotor_landing_amd.trainer as T
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())
    real = T.PromotionWindow

    class Scripted(real):
        resets = 0
        def reset(self):
            Scripted.resets += 1
            return super().reset()
        def push_flags(self, flags):
            super().push_flags(flags)
            return 0 if Scripted.resets <= 3 and len(flags) else None   # resets 1 (constructor's level-0 start), 2 (level 0), 3 (level 1): see the asserts below
    monkeypatch.setattr(T, "PromotionWindow", Scripted)
    kw = dict(curriculum_steps=3, n_envs=48, chunk_steps=8, checkpoint_every=10**9, max_num_episodes=600, t_max=3, mode="paper", judge_envs=16,
              successive_successful_episodes=10, eps_floor=0.3, restart_after=1.5, step_back_after=1, max_step_backs=2)
    tr = T.Trainer(save_path=tmp_path / "a", **kw)
    h = tr.curriculum_training()
    assert [x["level"] for x in h] == [0, 1, 2]
    assert h[0]["promoted"] and h[1]["promoted"] and not h[2]["promoted"] and h[2]["exhausted"]
    assert h[2]["step_backs"] == 2 and h[1]["step_backs"] >= 1          # level 1 was learnt again (twice) and fell back to its promoted entry each time
    assert h[1]["restarts"] == 0 and h[1]["wall_first_promoted_s"] is not None


def _table_score(tr):
    """a landing score the CPU can compute (the real one flies the HIP engine): a deterministic function of the final tables, different per attempt"""
    a = tr._double_q_learning_agent
    v = float(np.abs(a.Q_table_a).sum() + 3.0 * np.abs(a.Q_table_b).sum())
    return {"touchdown_rate": (v % 1000.0) / 1000.0, "goal_hold_rate": 0.5}


def _attempts_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import json
    import torch.distributed as dist
    from _torch_comm import TorchComm
    import dql_multirotor_landing_amd.trainer as T
    from dql_multirotor_landing_amd.attempts import attempt_seed, curriculum_attempts
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    T.Engine = _oracle_engine_class()
    comm = TorchComm()
    scored = []

    def score(tr):
        scored.append(rank)
        return _table_score(tr)
    res = curriculum_attempts(lambda j: T.Trainer(save_path=Path(out_dir) / f"run{j}", comm=comm, **{**TR_KW, "seed": attempt_seed(TR_KW.get("seed", 42), j)}), score,
                              max_attempts=3, accept_touchdown=2.0, comm=comm, rank=rank, close=lambda tr: None)
    assert scored == ([0, 0, 0] if rank == 0 else [])   # rank 0 alone scores; the other rank gets the figures through the all-reduce
    qa, qb, cnt = res["trainer"]._engine.get_tables()
    np.savez(Path(out_dir) / f"attempts_rank{rank}.npz", qa=qa, count=cnt)
    (Path(out_dir) / f"attempts{rank}.json").write_text(json.dumps({"chosen": res["chosen"], "accepted": res["accepted"], "selection": [a["selection"] for a in res["attempts"]],
                                                                      "promoted": [a["promoted_levels"] for a in res["attempts"]], "history": _strip(res["history"])}))
    dist.barrier()
    dist.destroy_process_group()


def test_curriculum_attempts_two_ranks_equal_single_process(tmp_path, monkeypatch):
    """Several whole curricula (attempts.py) with the real Trainer sharded over two ranks (gloo, oracle engine): both ranks choose the same attempt on rank 0's
    scores, and attempts, scores, choice and chosen tables are those of one process."""
    import json
    import torch.multiprocessing as mp
    import dql_multirotor_landing_amd.trainer as T
    from dql_multirotor_landing_amd.attempts import attempt_seed, curriculum_attempts
    mp.spawn(_attempts_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.loads((tmp_path / f"attempts{r}.json").read_text()) for r in range(2))
    assert r0 == r1 and len(r0["selection"]) == 3 and not r0["accepted"]
    assert len({s["touchdown_rate"] for s in r0["selection"]}) == 3   # three different runs
    best = max(range(3), key=lambda k: (r0["promoted"][k], r0["selection"][k]["touchdown_rate"], -k))
    assert r0["chosen"] == best
    monkeypatch.setattr(T, "Engine", _oracle_engine_class())
    one = curriculum_attempts(lambda j: T.Trainer(save_path=tmp_path / f"single{j}", **{**TR_KW, "seed": attempt_seed(TR_KW.get("seed", 42), j)}), _table_score,
                              max_attempts=3, accept_touchdown=2.0, close=lambda tr: None)
    assert one["chosen"] == r0["chosen"] and [a["selection"] for a in one["attempts"]] == r0["selection"]
    assert json.loads(json.dumps(_strip(one["history"]))) == r0["history"]
    qa, _, cnt = one["trainer"]._engine.get_tables()
    for r in range(2):
        z = np.load(tmp_path / f"attempts_rank{r}.npz")
        np.testing.assert_array_equal(z["qa"], qa)
        np.testing.assert_array_equal(z["count"], cnt)
