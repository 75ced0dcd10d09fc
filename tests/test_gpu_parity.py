"""HIP product vs the CPU oracle on the same seeded inputs (run with -m gpu on an MI355X).

Bar (BASELINE.json north_star): discrete state indices, actions and tables bit-exact; continuous dynamics within
1e-5 relative.  Because oracle and kernel spell out the same IEEE-754 operation sequence (explicit fma, no
contraction, own elementary functions), the same-dtype comparison is asserted EXACT for every field; the 1e-5 bound
is what the float32 kernel must hold against the float64 oracle over one agent period from identical states."""
import os
from pathlib import Path

import numpy as np
import pytest

from dql_multirotor_landing_amd.config import DqlConfig, F32, F64, TRAJ_EIGHT, Q_PAPER

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from dql_multirotor_landing_amd.engine import Engine
    from oracle.oracle import Oracle
    return Engine, Oracle


def _compare(eng, orc, exact=True, what=""):
    er, ei = eng.get_fields()
    o_r, o_i = orc.get_fields()
    names = eng.field_names(); inames = eng.field_names(True)
    assert names == orc.field_names() and inames == orc.field_names(True)
    for k, nm in enumerate(inames):
        assert np.array_equal(ei[k], o_i[k]), f"{what}: int field {nm} differs in {np.flatnonzero(ei[k] != o_i[k])[:5]}"
    for k, nm in enumerate(names):
        if exact:
            assert np.array_equal(er[k], o_r[k]), f"{what}: field {nm} max abs diff {np.abs(er[k] - o_r[k]).max()}"
        else:
            np.testing.assert_allclose(er[k], o_r[k], rtol=1e-5, atol=1e-6, err_msg=f"{what}: field {nm}")
    qa, qb, cnt = eng.get_tables()
    assert np.array_equal(qa.ravel(), orc.qa), f"{what}: Q_table_a"
    assert np.array_equal(qb.ravel(), orc.qb), f"{what}: Q_table_b"
    assert np.array_equal(cnt.ravel(), orc.count), f"{what}: state_action_counter"


@pytest.mark.parametrize("dtype", [F32, F64])
@pytest.mark.parametrize("n", [1, 257, 4096])
def test_training_bit_exact_vs_oracle(mods, dtype, n):
    Engine, Oracle = mods
    steps = 60 if n <= 257 else 25
    eng = Engine(DqlConfig(dtype=dtype), n, seed=42)
    orc = Oracle(DqlConfig(dtype=dtype), n, seed=42)
    for chunk, eps in ((steps // 2, 1.0), (steps - steps // 2, 0.1)):
        eng.train_steps(chunk, eps); orc.train_steps(chunk, eps)
        _compare(eng, orc, exact=True, what=f"n={n} dtype={dtype} eps={eps}")
    se, so = eng.stats(), orc.stats_dict()
    assert se["decisions"] == so["decisions"] and se["episodes"] == so["episodes"]
    assert list(se["by_code"].values()) == so["by_code"]
    assert se["reward_sum"] == so["reward_sum"]  # fixed-point sums: order independent


def test_long_run_with_episode_ends(mods):
    """600 agent periods: time-outs at step 459 (B18), fly-zone exits, resets, sticky success (B8)."""
    Engine, Oracle = mods
    n = 192
    eng = Engine(DqlConfig(dtype=F32), n, seed=3)
    orc = Oracle(DqlConfig(dtype=F32), n, seed=3)
    for _ in range(6):
        eng.train_steps(100, 0.3); orc.train_steps(100, 0.3)
        _compare(eng, orc, exact=True, what="long run")
    st = eng.stats()
    assert st["episodes"] > n  # every env finished at least one episode on average
    assert st["by_code"]["TERMINAL_TIMEOUT"] + st["by_code"]["TERMINAL_FLYZONE_X"] + st["by_code"]["TERMINAL_SUCCESS"] == st["episodes"]


@pytest.mark.parametrize("kw", [
    dict(working_curriculum_step=2), dict(working_curriculum_step=4, init_uniform=1, vz_setpoint=-0.4),
    dict(quirks=Q_PAPER), dict(trajectory=TRAJ_EIGHT), dict(per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1),
    dict(block=256), dict(block=128),
    dict(two_axis=1), dict(two_axis=1, working_curriculum_step=3, init_uniform=1), dict(two_axis=1, quirks=Q_PAPER, trajectory=TRAJ_EIGHT),
    dict(two_axis=1, goal_logic=0, vz_setpoint=-0.4, working_curriculum_step=4, init_uniform=1),
    # MDP constants that are NOT the reference's: the literal-table layouts (LitM) must not be selected — the constants buffer serves (round 5: dql_create's refm_matches)
    dict(p_max=4.0, init_sigma=4.0 / 3), dict(f_ag=20.0, working_curriculum_step=1), dict(p_max=5.0, init_sigma=5.0 / 3, two_axis=1, block=256),
])
def test_config_variants_bit_exact(mods, kw):
    Engine, Oracle = mods
    kw = dict(kw)
    block = kw.pop("block", 0)
    n = 320
    eng = Engine(DqlConfig(dtype=F32, **kw), n, seed=11)
    eng.set_option("block", block)
    orc = Oracle(DqlConfig(dtype=F32, **kw), n, seed=11)
    # start from the reference's stage-4 tables so that greedy actions and bootstraps are non-trivial
    from pathlib import Path
    g = Path(__file__).parent / "golden" / "assets"
    qa, qb, cnt = (np.load(g / f) for f in ("Q_table_a.npy", "Q_table_b.npy", "state_action_count.npy"))
    eng.set_tables(qa, qb, cnt)
    orc.set_tables(qa, qb, cnt)
    eng.train_steps(80, 0.2); orc.train_steps(80, 0.2)
    _compare(eng, orc, exact=True, what=str(kw))


def test_external_actions_and_eval(mods):
    Engine, Oracle = mods
    n = 128
    rng = np.random.default_rng(0)
    eng = Engine(DqlConfig(dtype=F64), n, seed=5)
    orc = Oracle(DqlConfig(dtype=F64), n, seed=5)
    for _ in range(40):
        a = rng.integers(0, 3, n).astype(np.uint8)
        eng.step(a); orc.step(a)
    _compare(eng, orc, exact=True, what="external actions")
    np.testing.assert_array_equal(eng.states(), orc.get_fields()[1][0])
    eng.eval_steps(20); orc.eval_steps(20)
    _compare(eng, orc, exact=True, what="greedy eval")
    with pytest.raises(ValueError):
        eng.step(np.full(n, 3, dtype=np.uint8))


from dql_multirotor_landing_amd.state_layout import to_f32_filter_state  # float64 histories -> float32 transposed filter states


F32_VS_F64_CASES = {
    "default": {},
    "configs4": dict(per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1),          # the headline flavour (BASELINE configs[4])
    "two_axis": dict(two_axis=1),                                                          # configs[2]
    "two_axis_configs4": dict(two_axis=1, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1),
}
_DYN = ["px", "py", "pz", "vx", "vy", "vz", "qw", "qx", "qy", "qz", "wx", "wy", "wz", "om0", "om1", "om2", "om3", "vz_i", "yw_i", "vz_state", "yw_state",
        "mp_phase", "mp_x", "mp_u", "mp_y", "mp_v", "pitch_sp", "roll_sp"]
_OBS = ["obs_p_x", "obs_v_x", "obs_a_x", "obs_p_y", "obs_v_y", "obs_a_y", "kal_x_x", "kal_y_x", "kal_x_P", "kal_y_P"]
_REW = ["reward", "cum_x", "cum_y", "shp_x_p", "shp_x_v", "shp_x_a", "shp_y_p", "shp_y_v", "shp_y_a"]


@pytest.mark.parametrize("periods", [1, 16])
@pytest.mark.parametrize("case", list(F32_VS_F64_CASES))
def test_f32_kernel_vs_f64_oracle(mods, case, periods):
    """north_star tolerance: continuous dynamics within 1e-5 relative — the float32 KERNEL (the arithmetic every throughput figure runs on, in its
    shortest forms) against the float64 ORACLE (the reference's expressions operation by operation, pinned bit for bit by G1-G13) from identical
    states, on the flavours the bench flies: default, the configs[4] flags (per-env platforms + observation noise + Kalman R > 0), two-axis, both.
    The common state is flown by the float64 oracle (50 periods) and handed to the float32 engine (filter states mapped: state_layout.py); then ONE
    launch of 1 or 16 agent periods on both.  Compared: every dynamic field, the latched observation + Kalman state, reward / shaping / cumulative
    reward, discrete indices.  Bounds (relative to max(|x|, 1); measured maxima at 4 096 envs in profiles/r5_f32_vs_f64_survey.jsonl):
      1 period:   dynamics + observation 1e-5 (measured <= 2.6e-6);  reward, shaping, cumulative: 2.5e-4 ABSOLUTE — they are the position / velocity error
                  times |w_p| / p_max = 22 (pkg/mdp.py:441-541) (measured 5.3e-5)
      16 periods: one launch, the env in registers throughout: errors add up over 350 physics ticks and the float32 platform phase drifts by half
                  an ulp per manager tick (fixture_checks.check_g11_platform): dynamics + observation 1e-4 (measured 4.2e-5), reward terms 5e-3 (1.7e-3)
    Envs whose episode ended in a different period in the two dtypes (a state within rounding of a bin edge or a fly-zone limit) are excluded and
    counted: < 0.5 %, as are differing discrete indices among the rest."""
    Engine, Oracle = mods
    kw = F32_VS_F64_CASES[case]
    n = 4096
    o64 = Oracle(DqlConfig(dtype=F64, **kw), n, seed=9, n_threads=8)
    o64.train_steps(50, 1.0)
    reals, ints = o64.get_fields()
    e32 = Engine(DqlConfig(dtype=F32, **kw), n, seed=9)
    e32.train_steps(50, 1.0)  # advance the schedule identically, then overwrite tables and env state
    names, inames = e32.field_names(), e32.field_names(True)
    qa, qb, cnt = o64.qa.copy(), o64.qb.copy(), o64.count.copy()
    e32.set_tables(qa, qb, cnt)  # master == acting on both sides from here
    o64.set_tables(qa, qb, cnt)
    e32.set_fields(to_f32_filter_state(reals, names), ints)
    e32.set_option("periods_per_launch", periods); o64.set_option("periods_per_launch", periods)
    e32.train_steps(periods, 1.0); o64.train_steps(periods, 1.0)
    r32, i32 = e32.get_fields(); r64, i64 = o64.get_fields()
    same = np.ones(n, dtype=bool)
    for k in ("step_count", "code", "flags", "cur_check"):
        same &= i32[inames.index(k)] == i64[inames.index(k)]
    assert same.mean() > 0.995, f"{(~same).sum()} envs ended their episode in different periods"
    tol_dyn, tol_rew = (1e-5, 2.5e-4) if periods == 1 else (1e-4, 5e-3)
    for k in _DYN + _OBS:
        j = names.index(k)
        err = (np.abs(r32[j] - r64[j]) / np.maximum(np.abs(r64[j]), 1.0))[same].max()
        assert err < tol_dyn, (k, err)
    for k in _REW:
        j = names.index(k)
        err = np.abs(r32[j] - r64[j])[same].max()
        assert err < tol_rew, (k, err)
    for ax in (0, 1):
        assert (i32[ax] != i64[ax])[same].mean() < 5e-3
    assert np.array_equal(i32[inames.index("action")][same], i64[inames.index("action")][same])  # eps = 1: the Philox stream, identical in both
    if "noise_pos_sd" in kw:  # the noise really is on the latched observation (and equal in both dtypes up to rounding)
        j = names.index("obs_p_x")
        assert np.abs(r64[j] - (r64[names.index("mp_x")] - r64[names.index("px")])).max() > 0.05
    if periods == 1:  # the mapped filter states are the same filters: after the period they still describe the float64 histories
        want = to_f32_filter_state(r64, names)
        for k in ("vz_x1", "vz_x2", "vz_y1", "yw_x1", "yw_x2", "yw_y1"):
            j = names.index(k)
            np.testing.assert_allclose(r32[j][same], want[j][same], rtol=2e-4, atol=2e-5, err_msg=k)


def test_kalman_fixed_point_shortcut_only_fires_on_the_devices_own_fixed_point(mods):
    """the float32 step skips the Kalman covariance update when every lane of the wave sits on P's fixed point (host: kalman_fixed_point,
    same three operations in float32).  Plant the fixed point in ONE lane of a wave whose other lanes are still converging, a neighbouring value
    (1 ulp off) in another, and compare with the oracle, which always runs the plain update: bit for bit, so the shortcut changed nothing."""
    Engine, Oracle = mods
    kw = dict(dtype=F32, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1)
    n = 128
    eng = Engine(DqlConfig(**kw), n, seed=4); orc = Oracle(DqlConfig(**kw), n, seed=4)
    names = eng.field_names()
    # the fixed point itself: fly one engine until P stops moving (1 s of simulated time is enough, B15: never reset)
    probe = Engine(DqlConfig(**kw), 64, seed=1)
    probe.train_steps(60, 1.0)
    p_fix = np.float32(probe.get_fields()[0][names.index("kal_x_P")][0])
    probe.train_steps(1, 1.0)
    assert np.float32(probe.get_fields()[0][names.index("kal_x_P")][0]) == p_fix and 0 < p_fix < 1
    reals, ints = eng.get_fields()
    j = names.index("kal_x_P")
    reals[j, 3] = p_fix                                              # on the fixed point from the start, alone in its wave
    reals[j, 70] = np.nextafter(p_fix, np.float32(1.0))              # 1 ulp off: must take the plain update
    reals[j, 64:128][1::2] = p_fix                                   # half a wave on it, half not
    eng.set_fields(reals, ints); orc.set_fields(reals, ints)
    for _ in range(3):
        eng.train_steps(16, 1.0); orc.train_steps(16, 1.0)
        _compare(eng, orc, exact=True, what="planted Kalman covariance")
    assert (np.float32(eng.get_fields()[0][j]) == p_fix).all()      # every lane has converged onto the same fixed point


def test_curriculum_switch_and_transfer(mods):
    Engine, Oracle = mods
    n = 96
    eng = Engine(DqlConfig(dtype=F32), n, seed=1)
    orc = Oracle(DqlConfig(dtype=F32), n, seed=1)
    eng.train_steps(30, 1.0); orc.train_steps(30, 1.0)
    eng.transfer(1, 0.8172650252856599); orc.transfer(1, 0.8172650252856599)
    eng.set_curriculum(1); orc.set_curriculum(1)
    eng.train_steps(30, 0.0); orc.train_steps(30, 0.0)
    _compare(eng, orc, exact=True, what="level 1")
    assert (eng.states() // 189).max() <= 1


def test_windowed_accumulation_matches_oracle(mods):
    """Multi-GPU semantics on one GPU: window accumulators, local work-table updates, fold into the base tables."""
    Engine, Oracle = mods
    n = 256
    eng = Engine(DqlConfig(dtype=F32), n, seed=21)
    orc = Oracle(DqlConfig(dtype=F32), n, seed=21)
    eng.set_windowed(True); orc.set_windowed(True)
    for _ in range(3):
        eng.train_steps(8, 0.5); orc.train_steps(8, 0.5)
        np.testing.assert_array_equal(eng.get_accum(), orc.get_accum())
        _compare(eng, orc, exact=True, what="inside window")
        eng.apply_accum(); orc.apply_accum()
        assert not eng.get_accum().any()
        _compare(eng, orc, exact=True, what="after fold")


_RCCL_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from dql_multirotor_landing_amd import _lib
from dql_multirotor_landing_amd.comm import RcclComm
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.dist import LocalWindowReducer, RcclWindowReducer, ShardedRunner
from dql_multirotor_landing_amd.engine import Engine
from oracle.oracle import Oracle
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
comm = RcclComm(rank=0, world=1, device=0)   # ncclGetUniqueId + ncclCommInitRank inside libdql_hip.so
assert "torch" not in sys.modules, "the RCCL path must not need PyTorch"
n = 512
eng = Engine(DqlConfig(dtype=F32), n, seed=33)
orc = Oracle(DqlConfig(dtype=F32), n, seed=33); orc.set_windowed(True)
run = ShardedRunner(eng, RcclWindowReducer(eng, comm), sync_period=4)
run.train_steps(12, 0.7)
for _ in range(3):
    orc.train_steps(4, 0.7); orc.flush(); orc.apply_accum()
eng.sync()
qa, qb, cnt = eng.get_tables()
er, ei = eng.get_fields(); o_r, o_i = orc.get_fields()
assert np.array_equal(qa.ravel(), orc.qa) and np.array_equal(cnt.ravel(), orc.count) and cnt.sum() > 0
assert np.array_equal(ei, o_i) and np.array_equal(er, o_r)
# exchange timing hooks
eng.kernel_timer(True); run.train_steps(8, 0.7); ms, k = eng.sync_time_ms(); eng.kernel_timer(False)
assert k == 2 and 0 < ms < 50, (ms, k)
# control plane through RCCL: sums, max, gathers on host arrays
v = comm.all_reduce_sum(np.array([1.5, -2.0, 3e9])); assert np.array_equal(v, [1.5, -2.0, 3e9])
v = comm.all_reduce_max(np.array([7.25])); assert v[0] == 7.25
v = comm.all_reduce_sum_i64(np.array([1 << 60, -5])); assert list(v) == [1 << 60, -5]
d = np.arange(6, dtype=np.uint64).reshape(2, 3) + np.uint64(1 << 63); g = d[::-1].copy()
gd, gg = comm.all_gather_masks(d, g); assert np.array_equal(gd, d) and np.array_equal(gg, g)
gd, gg = comm.all_gather_masks(d[:0], g[:0]); assert gd.shape == (0, 3)
comm.barrier()
# error paths: wrong device, detached context
try:
    eng2 = Engine(DqlConfig(dtype=F32), 64, seed=1); eng2.set_windowed(True); eng2.allreduce_window()
    raise SystemExit("allreduce without a communicator must fail")
except ValueError:
    pass
# the Trainer's control plane over the same communicator: counters summed and episode logs gathered by RCCL, table
# exchange through RcclWindowReducer; one rank, so the run must equal the local-exchange run (configs[3] flavour: the 2-level
# curriculum at 32 768 envs per rank with sync_period 2)
import json, tempfile
import dql_multirotor_landing_amd.trainer as T
kw = dict(n_envs=32768, mode="paper", chunk_steps=16, sync_period=2, max_num_episodes=40000, curriculum_steps=2, t_max=4,
          successive_successful_episodes=20, success_rate=0.2, judge_envs=300, checkpoint_every=10**9)
strip = lambda hist: [{k: v for k, v in h.items() if not k.startswith("wall")} for h in hist]
d = tempfile.mkdtemp()
a = T.Trainer(save_path=d + "/a", comm=comm, reducer_factory=comm.reducer, **kw)
ha = strip(a.curriculum_training())
b = T.Trainer(save_path=d + "/b", reducer_factory=LocalWindowReducer, **kw)
hb = strip(b.curriculum_training())
assert ha == hb and len(ha) == 2, (ha, hb)
for x, y in zip(a._engine.get_tables(), b._engine.get_tables()):
    assert np.array_equal(x, y)
a._engine.close(); b._engine.close()
# BASELINE configs[3] itself: the FULL 0 -> 4 curriculum at 32 768 envs per GPU with the unmodified 0.96 / 100 rule, the bench's trainer
# settings (bench.CURRICULUM_KW: 16 periods per launch, table exchange every 16, 2 judged envs, exploration tail) — through the RCCL
# reducer of one rank == through the local exchange: histories (promotions, episode counts, population success) and tables bit for bit
import bench
kw = dict(n_envs=32768, mode="paper", chunk_steps=64, sync_period=bench.CURRICULUM_SYNC, max_num_episodes=384 * 32768, checkpoint_every=10**9, seed=42, **bench.CURRICULUM_KW)
a = T.Trainer(save_path=d + "/c", comm=comm, reducer_factory=comm.reducer, **kw)
ha = strip(a.curriculum_training())
b = T.Trainer(save_path=d + "/d", reducer_factory=LocalWindowReducer, **kw)
hb = strip(b.curriculum_training())
assert ha == hb and [h["level"] for h in ha] == [0, 1, 2, 3, 4], (ha, hb)
assert ha[0]["promoted"] and ha[0]["success_rate"] > 0.95, ha[0]          # level 0 passes the reference's rule, and the POPULATION is there too
assert sum(h["promoted"] for h in ha) >= 4, ha
for x, y in zip(a._engine.get_tables(), b._engine.get_tables()):
    assert np.array_equal(x, y)
assert "torch" not in sys.modules
comm.close()
print("RCCL_OK")
"""


def test_rccl_reducer_world_size_1():
    """The RCCL exchange of libdql_hip.so on a single rank, in its own process and WITHOUT PyTorch: unique id, communicator,
    ncclAllReduce(int64, sum) of the window on the engine's stream, fold; the host-buffer control-plane collectives; the
    Trainer on the RCCL communicator.  With one rank the sum is the identity, so results equal the windowed oracle and
    the local-exchange Trainer — the 2-level miniature and the full 5-level curriculum at BASELINE configs[3]'s 32 768 envs per GPU."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_RCCL_RANK_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from dql_multirotor_landing_amd.comm import RcclComm
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.dist import RcclWindowReducer, ShardedRunner, shard_range
from dql_multirotor_landing_amd.engine import Engine
n_total, out = int(sys.argv[1]), sys.argv[2]
comm = RcclComm.from_env()                      # RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as a launcher exports them
lo, hi = shard_range(n_total, comm.rank, comm.world)
eng = Engine(DqlConfig(dtype=F32), hi - lo, seed=42, device=comm.device, env_id_offset=lo)
eng.set_option("periods_per_launch", 2)
run = ShardedRunner(eng, RcclWindowReducer(eng, comm), sync_period=2)
run.train_steps(40, 1.0); run.train_steps(40, 0.2); run.sync()
qa, qb, cnt = eng.get_tables()
reals, ints = eng.get_fields()
dec = comm.all_reduce_sum([float(eng.stats()["decisions"])])[0]
np.savez(f"{out}/rank{comm.rank}.npz", qa=qa, qb=qb, count=cnt, reals=reals, ints=ints, decisions=dec)
comm.barrier(); comm.close()
assert "torch" not in sys.modules
print("RANK_OK")
"""


def test_rccl_two_ranks_equal_one_process(tmp_path):
    """Two ranks on two GPUs, table exchange by ncclAllReduce(int64, sum) from inside libdql_hip.so every 2 agent periods ==
    one process with all the envs on the same schedule (the local exchange), bit for bit: tables, visit counts and every env's
    state (integer sums are order independent, the RNG is keyed by the global env id).  Needs 2 GPUs: skipped on a 1-GPU box,
    where tests/test_dist_gloo.py covers the same runner with a gloo stand-in communicator."""
    import ctypes as C
    import socket
    import subprocess
    import sys
    from dql_multirotor_landing_amd import _lib
    from dql_multirotor_landing_amd.dist import LocalWindowReducer, ShardedRunner
    from dql_multirotor_landing_amd.engine import Engine
    n = C.c_int(0)
    _lib.check(_lib.load().dql_device_count(C.byref(n)))
    if n.value < 2:
        pytest.skip("needs 2 GPUs")
    root = Path(__file__).resolve().parent.parent
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    n_total = 6000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _RCCL_RANK_SCRIPT, str(n_total), str(tmp_path)], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=600))
    finally:
        for pr in procs:  # exactly the processes started here
            if pr.poll() is None:
                pr.kill()
    for pr, (so, se) in zip(procs, outs):
        assert pr.returncode == 0 and "RANK_OK" in so, so[-1000:] + se[-3000:]
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    single = Engine(DqlConfig(dtype=F32), n_total, seed=42)
    single.set_option("periods_per_launch", 2)
    run = ShardedRunner(single, LocalWindowReducer(single), sync_period=2)
    run.train_steps(40, 1.0); run.train_steps(40, 0.2); run.sync()
    qa, qb, cnt = single.get_tables()
    for r in ranks:
        assert np.array_equal(r["qa"], qa) and np.array_equal(r["qb"], qb) and np.array_equal(r["count"], cnt)
        assert r["decisions"] == single.stats()["decisions"]
    reals, ints = single.get_fields()
    assert np.array_equal(np.concatenate([ranks[0]["ints"], ranks[1]["ints"]], axis=1), ints)
    assert np.array_equal(np.concatenate([ranks[0]["reals"], ranks[1]["reals"]], axis=1), reals)
    assert cnt.sum() > 0


_P2P_RANK_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.dist import P2PWindowReducer, ShardedRunner, shard_range
from dql_multirotor_landing_amd.engine import Engine
n_total, out, dev = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
lo, hi = shard_range(n_total, rank, world)
eng = Engine(DqlConfig(dtype=F32, t_max=5.0), hi - lo, seed=42, device=dev, env_id_offset=lo)
eng.set_option("periods_per_launch", 2)
run = ShardedRunner(eng, P2PWindowReducer(eng, rank, world), sync_period=2)
run.train_steps(40, 1.0); run.train_steps(41, 0.2); run.sync()
assert not eng.p2p_failed(), "a peer never showed up"
qa, qb, cnt = eng.get_tables()
reals, ints = eng.get_fields()
np.savez(f"{out}/rank{rank}.npz", qa=qa, qb=qb, count=cnt, reals=reals, ints=ints, decisions=eng.stats()["decisions"])
assert "torch" not in sys.modules
print("RANK_OK")
"""


@pytest.mark.parametrize("world", [2, 3, 5])  # 5 rank processes + this one: the GPU box admits 6 processes on its card
def test_p2p_exchange_ranks_sharing_one_gpu_equal_one_process(tmp_path, world):
    """The one-shot peer-to-peer exchange (dql_p2p_*: HIP IPC mappings of uncached exchange buffers, flags with system-scope
    release / acquire, slots summed in rank order) with `world` rank PROCESSES — all on GPU 0, which is what a 1-GPU box can host; on a
    node they would sit on different GPUs — against one process with all the envs on the same schedule: tables, visit counts and
    every env's state bit for bit.  Also the first test in which several HIP-engine ranks really run side by side."""
    import subprocess
    import sys
    from dql_multirotor_landing_amd.dist import LocalWindowReducer, ShardedRunner
    from dql_multirotor_landing_amd.engine import Engine
    root = Path(__file__).resolve().parent.parent
    n_total = 3000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), DQL_COMM_ID_FILE=str(tmp_path / "boot.id"), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _P2P_RANK_SCRIPT, str(n_total), str(tmp_path), "0"], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=300))
    finally:
        for pr in procs:  # exactly the processes started here
            if pr.poll() is None:
                pr.kill()
    for pr, (so, se) in zip(procs, outs):
        assert pr.returncode == 0 and "RANK_OK" in so, so[-1000:] + se[-3000:]
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    single = Engine(DqlConfig(dtype=F32, t_max=5.0), n_total, seed=42)
    single.set_option("periods_per_launch", 2)
    run = ShardedRunner(single, LocalWindowReducer(single), sync_period=2)
    run.train_steps(40, 1.0); run.train_steps(41, 0.2); run.sync()
    qa, qb, cnt = single.get_tables()
    for r in ranks:
        assert np.array_equal(r["qa"], qa) and np.array_equal(r["qb"], qb) and np.array_equal(r["count"], cnt)
    assert sum(int(r["decisions"]) for r in ranks) == single.stats()["decisions"]
    reals, ints = single.get_fields()
    assert np.array_equal(np.concatenate([r["ints"] for r in ranks], axis=1), ints)
    assert np.array_equal(np.concatenate([r["reals"] for r in ranks], axis=1), reals)
    assert cnt.sum() > 0


_P2P_LONELY_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.dist import P2PWindowReducer
from dql_multirotor_landing_amd.engine import Engine
rank = int(os.environ["RANK"])
eng = Engine(DqlConfig(dtype=F32), 256, seed=1, env_id_offset=256 * rank)
red = P2PWindowReducer(eng, rank, 2)
eng.set_windowed(True)
if rank == 0:
    eng.set_option("p2p_spin_limit", 200000)
    eng.train_steps(4, 1.0)
    own = eng.get_accum()   # flushes: the window now holds this rank's own sums
    t0 = time.time(); red.all_reduce(); eng.sync(); dt = time.time() - t0
    assert eng.p2p_failed() and eng.p2p_failed_seq() == 1, "the wait must give up when the peer never pushes, and say which exchange"
    import numpy as np
    assert np.array_equal(eng.get_accum(), own), "a given-up exchange must leave the window untouched (this rank's own sums), not half-summed"
    try:  # the training loop's per-chunk synchronisation point: the run must END here, not carry on with diverging replicas
        eng.stats()
        raise SystemExit("stats() did not raise after a failed exchange")
    except RuntimeError as e:
        assert "gave up waiting for a peer" in str(e), str(e)
    print("GAVE_UP %.2f" % dt)
else:
    time.sleep(6)   # connected, but never takes part in the exchange
    print("IDLE_OK")
"""


def test_p2p_exchange_gives_up_on_a_missing_peer(tmp_path):
    """A rank whose peer never pushes: the wait kernel's poll loop is bounded (option "p2p_spin_limit"), every wave exits, the failure is
    reported by dql_p2p_status (with the number of the failed exchange) — no hang, no GPU reset; ONE waiter decides, so the window is
    untouched rather than half-summed; and `stats()`, the Trainer's per-chunk synchronisation point, raises from then on."""
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", DQL_COMM_ID_FILE=str(tmp_path / "boot.id"), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _P2P_LONELY_SCRIPT], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=120))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    assert procs[0].returncode == 0 and "GAVE_UP" in outs[0][0], outs[0][0][-500:] + outs[0][1][-2000:]
    assert procs[1].returncode == 0 and "IDLE_OK" in outs[1][0], outs[1][1][-2000:]
    assert float(outs[0][0].split("GAVE_UP")[1].split()[0]) < 5.0


def test_p2p_group_of_eight_ranks_in_one_process_equals_one_engine(mods):
    """world = DQL_P2P_MAX_RANKS: eight ranks x 512 envs driven by ONE host thread (dist.ShardedGroup, buffers connected by pointer:
    dql_p2p_connect_local; pushes of all ranks enqueued before any wait) on GPU 0 against one engine of 4 096 envs on the same windowed
    schedule — slot layout [2 parities][8 ranks], the full flag array, parity flips over 41 exchanges: tables, counts, every env."""
    from dql_multirotor_landing_amd.dist import LocalWindowReducer, ShardedGroup, ShardedRunner, shard_range
    from dql_multirotor_landing_amd.engine import Engine
    world, n_total = 8, 4096
    engs = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        e = Engine(DqlConfig(dtype=F32, t_max=5.0), hi - lo, seed=42, env_id_offset=lo)
        e.set_option("periods_per_launch", 2)
        engs.append(e)
    grp = ShardedGroup(engs, sync_period=2)
    grp.train_steps(40, 1.0); grp.train_steps(41, 0.2); grp.sync()
    single = Engine(DqlConfig(dtype=F32, t_max=5.0), n_total, seed=42)
    single.set_option("periods_per_launch", 2)
    run = ShardedRunner(single, LocalWindowReducer(single), sync_period=2)
    run.train_steps(40, 1.0); run.train_steps(41, 0.2); run.sync()
    qa, qb, cnt = single.get_tables()
    assert cnt.sum() > 0
    for e in engs:
        assert not e.p2p_failed()
        a, b, c = e.get_tables()
        assert np.array_equal(a, qa) and np.array_equal(b, qb) and np.array_equal(c, cnt)
    reals, ints = single.get_fields()
    parts = [e.get_fields() for e in engs]
    assert np.array_equal(np.concatenate([p[1] for p in parts], axis=1), ints)
    assert np.array_equal(np.concatenate([p[0] for p in parts], axis=1), reals)
    assert sum(e.stats()["decisions"] for e in engs) == single.stats()["decisions"]
    for e in engs:
        e.close()
    single.close()


@pytest.mark.parametrize("P,sync", [(1, None), (2, 4), (8, 8)])
def test_engine_resume_equals_uninterrupted_run(mods, P, sync):
    """get_fields + step_index -> fresh Engine -> set_tables + set_fields + set_step_index -> continue == the uninterrupted
    run bit for bit (tables, every env, statistics), on the HIP engine, with several periods per launch (the ping-pong buffers follow
    the launch parity, the tick schedule and the RNG counters the period index) and on the windowed schedule (base tables)."""
    from dql_multirotor_landing_amd.dist import LocalWindowReducer, ShardedRunner
    from dql_multirotor_landing_amd.engine import Engine
    cfg = dict(dtype=F32, t_max=4.0, fold_per_step=1)
    n = 1500

    def make():
        e = Engine(DqlConfig(**cfg), n, seed=9, env_id_offset=77)
        e.set_option("periods_per_launch", P)
        return e, ShardedRunner(e, LocalWindowReducer(e) if sync else None, sync_period=sync or 1)

    k1, k2 = 40 + (3 if not sync else 0), 56   # (an odd number of launches before the checkpoint when the schedule allows it)
    full, run = make()
    run.train_steps(k1, 0.7); run.sync()
    full.publish_tables()                      # a checkpoint is a table barrier in BOTH runs (Trainer.save)
    s_mid = full.stats()
    run.train_steps(k2, 0.3); run.sync()

    part, run_p = make()
    run_p.train_steps(k1, 0.7); run_p.sync()
    part.publish_tables()
    tabs, (reals, ints), j = part.get_tables(), part.get_fields(), part.step_index()
    assert j == k1
    part.close()
    back, run_b = make()
    back.set_tables(*tabs); back.set_fields(reals, ints); back.set_step_index(j)
    assert back.step_index() == j and back.stats()["agent_steps"] == 0   # the statistics restart, the period index does not
    run_b.train_steps(k2, 0.3); run_b.sync()
    for a, b in zip(back.get_tables(), full.get_tables()):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(back.get_fields(), full.get_fields()):
        np.testing.assert_array_equal(a, b)
    s_full, s_back = full.stats(), back.stats()
    assert s_back["agent_steps"] == k2 and s_full["agent_steps"] == k1 + k2
    for key in ("decisions", "episodes"):
        assert s_back[key] == s_full[key] - s_mid[key]
    assert s_back["by_code"] == {c: s_full["by_code"][c] - s_mid["by_code"][c] for c in s_full["by_code"]}
    full.close(); back.close()


def test_step_dev_equals_step(mods):
    """dql_step_dev (actions already in device memory, e.g. written by the caller's own policy kernel) == dql_step with the same actions
    from the host: states, rewards, dones, every field."""
    import ctypes as C
    from dql_multirotor_landing_amd.engine import Engine
    hip = C.CDLL("libamdhip64.so")
    n = 777
    a, b = Engine(DqlConfig(dtype=F32, t_max=3.0), n, seed=3), Engine(DqlConfig(dtype=F32, t_max=3.0), n, seed=3)
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(n)) == 0
    rng = np.random.default_rng(0)
    try:
        for _ in range(90):
            act = rng.integers(0, 3, n).astype(np.uint8)
            a.step(act)
            assert hip.hipMemcpy(dptr, act.ctypes.data_as(C.c_void_p), C.c_size_t(n), 1) == 0  # hipMemcpyHostToDevice
            b.step_dev(dptr.value)
        for x, y in zip(a.get_fields(), b.get_fields()):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a.rewards(), b.rewards())
        assert a.stats()["episodes"] == b.stats()["episodes"] > 0
    finally:
        b.sync(); hip.hipFree(dptr)
        a.close(); b.close()


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` starts N ranks itself; on a box with fewer GPUs it must fail loudly, never print n_gpus: 1."""
    import subprocess
    import sys
    from pathlib import Path
    from dql_multirotor_landing_amd import _lib
    import ctypes as C
    n = C.c_int(0)
    _lib.check(_lib.load().dql_device_count(C.byref(n)))
    want = n.value + 1
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(want), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-curriculum",
                        "--large-envs", "0"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0, r.stdout[-2000:]
    assert '"n_gpus"' not in r.stdout and "GPU" in r.stderr, r.stderr[-3000:]


def test_two_axis_bit_exact_f64_and_external_actions(mods):
    """BASELINE configs[2]: joint x+y MDP (two 1-D MDPs per env on shared tables, roll channel flown)."""
    Engine, Oracle = mods
    n = 384
    eng = Engine(DqlConfig(dtype=F64, two_axis=1), n, seed=17)
    orc = Oracle(DqlConfig(dtype=F64, two_axis=1), n, seed=17)
    eng.train_steps(120, 0.6); orc.train_steps(120, 0.6)
    _compare(eng, orc, exact=True, what="two axis f64")
    st = eng.stats()
    assert st["by_code"]["TERMINAL_FLYZONE_Y"] > 0, "the y axis is really flown"
    rng = np.random.default_rng(2)
    for _ in range(30):
        a = (rng.integers(0, 3, n) | (rng.integers(0, 3, n) << 2)).astype(np.uint8)
        eng.step(a); orc.step(a)
    _compare(eng, orc, exact=True, what="two axis external actions")
    assert eng.state_bytes_per_env() > Engine(DqlConfig(dtype=F64), 4, seed=1).state_bytes_per_env()


@pytest.mark.parametrize("n,block", [(1000, 0), (1000, 256), (8200, 0)])
def test_episode_log_matches_oracle(mods, n, block):
    """Completion order for the promotion rule (pkg/trainer.py:218-232): per agent period and wave of 64 envs, which envs
    finished an episode and which of those reached the goal state; n not a multiple of 64; read empties the log; a full log
    refuses the next launch."""
    Engine, Oracle = mods
    cfg = dict(dtype=F32, t_max=3.0)
    eng = Engine(DqlConfig(**cfg), n, seed=11)
    orc = Oracle(DqlConfig(**cfg), n, seed=11, n_threads=8)
    if block:
        eng.set_option("block", block)
    eng.episode_log_enable(40); orc.episode_log_enable(40)
    tot = np.zeros(2, dtype=np.int64)
    for eps in (1.0, 0.2, 0.2):
        eng.train_steps(40, eps); orc.train_steps(40, eps)
        de, ge = eng.episode_log_read()
        do, go = orc.episode_log_read()
        assert de.shape == (40, (n + 63) // 64) and de.dtype == np.uint64
        assert np.array_equal(de, do) and np.array_equal(ge, go)
        assert not (ge & ~de).any()
        tot += [int(np.bitwise_count(de).sum()), int(np.bitwise_count(ge).sum())]
    st = eng.stats()
    assert tot[0] == st["episodes"] > 0 and tot[1] == st["by_code"]["TERMINAL_SUCCESS"] > 0
    d2, _ = eng.episode_log_read()
    assert d2.shape[0] == 0
    eng.train_steps(40, 0.2)
    with pytest.raises(ValueError):
        eng.train_steps(1, 0.2)  # DQL_ESTATE: log full
    eng.episode_log_read()
    # restricted read (dql_episode_log_read_words): the first k words of every period, same bits, log emptied all the same
    twin = Engine(DqlConfig(**cfg), n, seed=11)
    if block:
        twin.set_option("block", block)
    twin.episode_log_enable(40)
    for eps in (1.0, 0.2, 0.2, 0.2):  # the same launches eng has made so far
        twin.train_steps(40, eps); twin.episode_log_read(words=0)
    for k in (1, 3, 0, 10**6):
        eng.train_steps(40, 0.2); twin.train_steps(40, 0.2)
        df, gf = eng.episode_log_read()
        dk, gk = twin.episode_log_read(words=k)
        kk = min(k, (n + 63) // 64)
        assert dk.shape == (40, kk) and np.array_equal(dk, df[:, :kk]) and np.array_equal(gk, gf[:, :kk])
        assert twin.episode_log_read()[0].shape[0] == 0
    twin.close()
    eng.episode_log_enable(0)
    eng.train_steps(3, 0.2)


def test_double_q_learning_updates_both_tables_in_paper_mode(mods):
    """Without the reference's table-a-only quirk (B1/B2) the coin-picked table learns, valued by the other one (oracle:
    env_agent_period); with it, Q_table_b never changes.  Both tables and the shared counter bit-exact, windowed too."""
    Engine, Oracle = mods
    n = 640
    for quirks, expect_b in ((Q_PAPER, True), (0x7F, False)):
        eng = Engine(DqlConfig(dtype=F32, quirks=quirks), n, seed=5)
        orc = Oracle(DqlConfig(dtype=F32, quirks=quirks), n, seed=5, n_threads=8)
        eng.train_steps(150, 0.5); orc.train_steps(150, 0.5)
        _compare(eng, orc, exact=True, what=f"quirks {quirks:#x}")
        qa, qb, cnt = eng.get_tables()
        assert (qa != 0).any() and bool((qb != 0).any()) == expect_b
        if expect_b:  # roughly half of the updates each
            assert 0.3 < (qb != 0).sum() / max(1, (qa != 0).sum()) < 3.0
    eng = Engine(DqlConfig(dtype=F32, quirks=Q_PAPER), n, seed=6); eng.set_windowed(True)
    orc = Oracle(DqlConfig(dtype=F32, quirks=Q_PAPER), n, seed=6, n_threads=8); orc.set_windowed(True)
    for _ in range(5):
        eng.train_steps(7, 0.3); orc.train_steps(7, 0.3)
        acc = eng.get_accum()
        assert acc.shape == (4 * 2835,) and np.array_equal(acc, orc.get_accum()) and acc[2 * 2835:].any()
        eng.apply_accum(); orc.flush(); orc.apply_accum()
        _compare(eng, orc, exact=True, what="windowed double Q")


@pytest.mark.parametrize("tick,block,kw", [(1, 0, {}), (2, 0, {}), (2, 256, {}), (3, 64, {}), (3, 256, {}), (4, 0, {}), (4, 64, {}), (4, 512, {}), (1, 512, {}),
                                           (0, 512, dict(two_axis=1, trajectory=TRAJ_EIGHT)), (4, 512, dict(vz_setpoint=-0.4, working_curriculum_step=3, init_uniform=1)),
                                           (3, 0, dict(two_axis=1, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1)),
                                           # the bench's headline instance k_step<float,256,LIT> in TRAIN mode, 16 periods per launch: plain, and with the configs[4] flags + the Trainer's update rule
                                           (4, 256, dict(ppl=16)),
                                           (4, 256, dict(ppl=16, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1, quirks=Q_PAPER, fold_per_step=1))])
def test_tick_layouts_bit_exact(mods, tick, block, kw):
    """Options "tick" / "block": every layout of the 500 Hz loop (plain, VGPR constants, packed float32, literal constants) and every
    workgroup size (64 .. 512 threads, the last with the register budget of 4 waves per SIMD: cold values in scratch) computes the
    same bits as the oracle.  n spans full and ragged workgroups; 3 periods per launch; episodes end inside launches."""
    Engine, Oracle = mods
    n = 1100
    kw = dict(kw)
    ppl = kw.pop("ppl", 3)
    cfg = dict(dtype=F32, t_max=4.0, **kw)
    eng = Engine(DqlConfig(**cfg), n, seed=21); orc = Oracle(DqlConfig(**cfg), n, seed=21, n_threads=8)
    eng.set_option("tick", tick); eng.set_option("block", block)
    eng.set_option("periods_per_launch", ppl); orc.set_option("periods_per_launch", ppl)
    g = Path(__file__).parent / "golden" / "assets"
    qa, qb, cnt = (np.load(g / f) for f in ("Q_table_a.npy", "Q_table_b.npy", "state_action_count.npy"))
    eng.set_tables(qa, qb, cnt); orc.set_tables(qa, qb, cnt)
    for steps, eps in ((100, 0.3), (50, 0.0)):
        eng.train_steps(steps, eps); orc.train_steps(steps, eps)
        _compare(eng, orc, exact=True, what=f"tick={tick} block={block} {kw}")
    assert eng.stats()["episodes"] > n // 2


def test_literal_tick_needs_reference_vehicle(mods):
    """tick 4 is compiled with the reference vehicle's constants as literals (csrc/dql_refk.inc): a context whose constants differ
    in any bit refuses the option, and its automatic choice falls back to run-time constants — checked against the oracle."""
    Engine, Oracle = mods
    cfg = dict(dtype=F32, mass=0.75, t_max=4.0)
    eng = Engine(DqlConfig(**cfg), 700, seed=2); orc = Oracle(DqlConfig(**cfg), 700, seed=2)
    with pytest.raises(ValueError):
        eng.set_option("tick", 4)
    eng.set_option("block", 512)
    eng.train_steps(60, 0.5); orc.train_steps(60, 0.5)
    _compare(eng, orc, exact=True, what="modified vehicle, block 512")
    e64 = Engine(DqlConfig(dtype=F64), 64, seed=2)
    with pytest.raises(ValueError):
        e64.set_option("block", 512)
    with pytest.raises(ValueError):
        e64.set_option("tick", 4)
    Engine(DqlConfig(dtype=F32), 64, seed=2).set_option("tick", 4)  # the default config IS the reference vehicle


@pytest.mark.parametrize("P,n,block,kw", [(2, 4096, 0, {}), (3, 300, 64, dict(quirks=Q_PAPER, fold_per_step=1)), (4, 9000, 256, dict(t_max=3.0)),
                                          (2, 20000, 0, dict(two_axis=1, quirks=Q_PAPER)), (4, 70, 0, dict(dtype=F64, t_max=2.0, fold_per_step=1)),
                                          (8, 4096, 0, dict(t_max=3.0)), (8, 9000, 512, dict(fold_per_step=1, quirks=Q_PAPER)), (7, 300, 64, dict(dtype=F64, two_axis=1, t_max=2.0)),
                                          (16, 4096, 0, dict(t_max=3.0, fold_per_step=1)), (16, 9000, 256, dict(per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1, quirks=Q_PAPER)),
                                          (13, 300, 64, dict(dtype=F64, t_max=2.0)), (32, 4096, 0, dict(t_max=3.0, fold_per_step=1)),
                                          (32, 9000, 256, dict(per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1, quirks=Q_PAPER)), (24, 200, 64, dict(dtype=F64, two_axis=1, t_max=2.0))])
def test_periods_per_launch_bit_exact(mods, P, n, block, kw):
    """Option "periods_per_launch": P agent periods per kernel launch with the env in registers in between (one state round trip,
    one launch boundary, one table publication per P periods).  Same per-period arithmetic, RNG counters and tick schedule; the
    launch's tables act for all P periods and its accumulators collect all of them — the oracle groups its periods the same way.
    Step counts that are not multiples of P, resets inside a launch (short episodes), the episode log's P rows per launch."""
    Engine, Oracle = mods
    kw = dict(kw); dtype = kw.pop("dtype", F32)
    eng = Engine(DqlConfig(dtype=dtype, **kw), n, seed=11); orc = Oracle(DqlConfig(dtype=dtype, **kw), n, seed=11)
    eng.set_option("periods_per_launch", P); orc.set_option("periods_per_launch", P)
    eng.set_option("block", block)
    cap = max(128, 10 * P + 8)
    eng.episode_log_enable(cap); orc.episode_log_enable(cap)
    for steps, eps in ((P * 5 + 1, 1.0), (7, 0.3), (P * 9 + (P - 1), 0.0)):
        eng.train_steps(steps, eps); orc.train_steps(steps, eps)
        _compare(eng, orc, exact=True, what=f"P={P} steps={steps}")
        d1, g1 = eng.episode_log_read(); d2, g2 = orc.episode_log_read()
        assert d1.shape[0] == steps and np.array_equal(d1, d2) and np.array_equal(g1, g2)
    se, so = eng.stats(), orc.stats_dict()
    assert se["decisions"] == so["decisions"] and se["episodes"] == so["episodes"] and se["reward_sum"] == so["reward_sum"]
    assert se["agent_steps"] == sum((P * 5 + 1, 7, P * 9 + (P - 1)))
    eng.eval_steps(2 * P + 1); orc.eval_steps(2 * P + 1)
    _compare(eng, orc, exact=True, what="eval")
    # windowed exchange on top: the window counts agent periods, the fold takes min(visits, periods) learning-rate steps
    eng.set_windowed(True); orc.set_windowed(True)
    eng.train_steps(2 * P, 0.5); orc.train_steps(2 * P, 0.5)
    eng.flush(); orc.flush(); eng.apply_accum(); orc.apply_accum()
    _compare(eng, orc, exact=True, what="windowed")
    with pytest.raises(ValueError):
        eng.set_option("periods_per_launch", 33)


def test_float32_preconditions_are_checked_not_assumed(mods):
    """(a) the float32 tick clamps w^2 at rotor_max^2 before the root — exact only when that square is a float32, so a vehicle for which it
    is not is refused in float32 (and flies in float64); (b) an x-axis float32 context flies the attitude law's closed form for roll_sp == 0 and
    never reads the field: a non-zero roll_sp handed to it is refused instead of ignored (a two-axis or float64 context takes it)."""
    Engine, Oracle = mods
    with pytest.raises(ValueError, match="rotor_max"):
        Engine(DqlConfig(dtype=F32, rotor_max=800.1), 64)
    Engine(DqlConfig(dtype=F32, rotor_max=838.5), 64).close()    # 838.5^2 = 703 082.25 is a float32
    Engine(DqlConfig(dtype=F64, rotor_max=800.1), 64).close()
    for kw, ok in ((dict(dtype=F32), False), (dict(dtype=F32, two_axis=1), True), (dict(dtype=F64), True)):
        e = Engine(DqlConfig(**kw), 64, seed=1)
        reals, ints = e.get_fields()
        reals[e.field_names().index("roll_sp"), 5] = 0.05
        if ok:
            e.set_fields(reals, ints)
        else:
            with pytest.raises(ValueError, match="roll_sp"):
                e.set_fields(reals, ints)
        e.close()


def test_issue_priority_alternation_changes_no_result(mods):
    """Round 5: the waves that share a SIMD take turns at the issue priority (s_setprio by period + wave-slot parity; automatic when a context has more env
    waves than the device SIMDs, option "fair_prio" 0 / 1 to force).  Scheduling only: every field and table stays bit-identical to the oracle with it forced
    on in a small context (plain and literal-constant layouts, several waves per workgroup) and forced off."""
    Engine, Oracle = mods
    kw = dict(dtype=F32, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1)
    for tick, fair in ((4, 1), (1, 1), (4, 0)):
        eng = Engine(DqlConfig(**kw), 1500, seed=6); orc = Oracle(DqlConfig(**kw), 1500, seed=6)
        eng.set_option("block", 256); eng.set_option("tick", tick); eng.set_option("fair_prio", fair)
        eng.set_option("periods_per_launch", 5); orc.set_option("periods_per_launch", 5)
        eng.train_steps(37, 0.4); orc.train_steps(37, 0.4)
        _compare(eng, orc, exact=True, what=f"tick {tick} fair_prio {fair}")
        eng.close()
    e = Engine(DqlConfig(**kw), 64, seed=1)
    with pytest.raises(ValueError):
        e.set_option("fair_prio", 2)
