"""The CPU oracle against the golden vectors captured from the reference's own Python modules
(tests/golden/make_golden.py).  fp64 oracle: bit-exact for everything the reference computes in
numpy float64 (MDP, agent); tight tolerances where the reference arithmetic is third-party
(tf.transformations stand-ins) or where the oracle uses its own elementary functions."""
import json

import numpy as np
import pytest

from dql_multirotor_landing_amd.config import DqlConfig, F32, F64, N_CELLS, TRAJ_EIGHT
from oracle import oracle as orc

RATIOS = [1.0, 0.8172650252856599, 0.8211253690681617, 0.8257273369742982, 0.8311571820651724]


def unpack(idx):
    idx = np.asarray(idx)
    return np.stack([idx // 189, (idx // 63) % 3, (idx // 21) % 3, (idx // 7) % 3, idx % 7], axis=-1)


def pack(s):
    s = np.asarray(s).astype(np.int64)
    return ((((s[..., 0] * 3 + s[..., 1]) * 3 + s[..., 2]) * 3 + s[..., 3]) * 7 + s[..., 4]).astype(np.int32)


@pytest.mark.parametrize("level", range(5))
def test_g1_discretise_bit_exact(golden_dir, level):
    g = np.load(golden_dir / "g1_discretise.npz")
    x = g[f"in_{level}"]
    cfg = DqlConfig(working_curriculum_step=level, dtype=F64)
    idx = orc.discretise(cfg, x[:, 0], x[:, 1], x[:, 2], x[:, 3])
    assert (idx >= 0).all()
    np.testing.assert_array_equal(unpack(idx), g[f"state_{level}"])


@pytest.mark.parametrize("level", range(5))
def test_g1_discretise_f32_statistical(golden_dir, level):
    """float32 arithmetic only differs on inputs within rounding distance of a bin edge."""
    g = np.load(golden_dir / "g1_discretise.npz")
    x = g[f"in_{level}"]
    cfg = DqlConfig(working_curriculum_step=level, dtype=F32)
    idx = orc.discretise(cfg, x[:, 0], x[:, 1], x[:, 2], x[:, 3])
    mism = (unpack(idx) != g[f"state_{level}"]).any(axis=1)
    # the edge cases were placed within 1e-7 relative of the edges on purpose; random inputs must agree
    assert mism[:1500].mean() < 2e-3


@pytest.mark.parametrize("level", range(5))
def test_g2_traces_bit_exact(golden_dir, level):
    """check / reward / continuous_action / reset incl. quirks B7-B11, B18 through one persistent MDP per level."""
    t = np.load(golden_dir / "g2_traces.npz")[f"trace_{level}"]
    cfg = DqlConfig(working_curriculum_step=level, dtype=F64)
    ms = np.zeros((8, 1))
    ms[7] = 8
    prev = np.array([-1], dtype=np.int32)
    n_steps = 0
    for row in t:
        op, act = int(row[0]), int(row[1])
        obs = np.array([row[2], row[3], row[4], row[5], row[6], row[7], row[8]]).reshape(7, 1)
        if op == 0:
            # TrainingMdp.reset: everything but the shaping memory (B9), then the first discrete_state
            ms[0] = 0.0; ms[4] = 0.0; ms[5] = 0; ms[6] = 0; ms[7] = 8
            idx = orc.discretise(cfg, obs[0], obs[2], obs[3], obs[4])
            np.testing.assert_array_equal(unpack(idx)[0], row[9:14].astype(int))
            prev = idx.astype(np.int32)
            continue
        ms, idx, rew, done = orc.mdp_transition(cfg, [act], obs, ms, prev)
        np.testing.assert_array_equal(unpack(idx)[0], row[9:14].astype(int))
        assert int(ms[7, 0]) == int(row[14]), "check code"
        assert rew[0] == row[15], f"reward {rew[0]!r} != {row[15]!r}"
        assert int(done[0]) == int(row[16])
        assert ms[0, 0] == row[17], "pitch set-point (B11 float accumulator)"
        assert ms[4, 0] == row[18], "cumulative reward"
        assert int(ms[5, 0]) == int(row[19]) and int(ms[6, 0]) == int(row[20])
        prev = idx.astype(np.int32)
        n_steps += 1
    assert n_steps > 1000


def test_g4_agent_update_sequence_bit_exact(golden_dir):
    g = np.load(golden_dir / "g4_agent.npz")
    qa = np.zeros(N_CELLS); qb = np.zeros(N_CELLS); cnt = np.zeros(N_CELLS)
    sa = g["upd_sa"]
    cell = pack(sa[:, :5]) * 3 + sa[:, 5]
    ns = pack(g["upd_ns"])
    orc.agent_update(qa, qb, cnt, cell, ns, g["upd_alpha"], 0.99, g["upd_reward"])
    np.testing.assert_array_equal(qa, g["upd_Qa"].ravel())
    np.testing.assert_array_equal(qb, g["upd_Qb"].ravel())  # B1: table b is never written
    np.testing.assert_array_equal(cnt, g["upd_count"].ravel())
    assert not qb.any()


def test_g4_predict_all_states(golden_dir):
    g = np.load(golden_dir / "g4_agent.npz")
    qa = np.load(golden_dir / "assets" / "Q_table_a.npy")
    qb = np.load(golden_dir / "assets" / "Q_table_b.npy")
    idx = pack(g["predict_states"])
    np.testing.assert_array_equal(idx, np.arange(945))
    np.testing.assert_array_equal(orc.agent_predict(qa, qb, idx), g["predict_actions"])
    np.testing.assert_array_equal(orc.agent_predict(g["upd_Qa"], g["upd_Qb"], idx), g["predict_actions_scripted"])


def test_g4_transfer_incl_wrap(golden_dir):
    g = np.load(golden_dir / "g4_agent.npz")
    qa = g["tl_Qa_in"].ravel().copy(); qb = g["tl_Qb_in"].ravel().copy()
    for k in range(5):
        orc.transfer(qa, qb, k, RATIOS[k])
        np.testing.assert_array_equal(qa, g[f"tl_Qa_after{k}"].ravel())
        np.testing.assert_array_equal(qb, g[f"tl_Qb_after{k}"].ravel())


def test_g5_alpha_table(golden_dir):
    g = np.load(golden_dir / "g5_schedules.npz")
    tab = DqlConfig().alpha_table(1536)
    np.testing.assert_array_equal(tab, g["alphas"][:1536])
    assert (g["alphas"][1536:] == 0.02949).all()


def test_g8_butterworth_kalman_pid(golden_dir):
    g = np.load(golden_dir / "g8_filters.npz")
    np.testing.assert_array_equal(orc.butterworth_run(g["bw_in"]), g["bw_out"])
    for tag, sd in (("r0", 0.0), ("r01", 0.1)):
        vel = g[f"kf_vel_{tag}"]
        flags = np.array([(i % 17 == 0) for i in range(len(vel))], dtype=np.uint8)
        np.testing.assert_array_equal(orc.kalman_run(vel, flags, 1e-4, sd), g[f"kf_acc_{tag}"])
    for tag in ("vz", "yaw", "kd"):
        eff, integ = orc.pid_run(g[f"pid_{tag}_params"], g[f"pid_{tag}_state"])
        np.testing.assert_array_equal(integ, g[f"pid_{tag}_integral"])
        np.testing.assert_array_equal(eff, g[f"pid_{tag}_effort"])


def test_g9_attitude(golden_dir):
    """Allocation inverse is closed-form here (np.linalg.inv in the reference); rotation helpers are third-party."""
    g = np.load(golden_dir / "g9_attitude.npz")
    cfg = DqlConfig(dtype=F64)
    ia, ib, ic = 1 / (4 * cfg.k_f), 1 / (2 * cfg.arm_length * cfg.k_f), 1 / (4 * cfg.k_f * cfg.k_m)
    Ainv = np.array([[0, -ib, ic, ia], [ib, 0, -ic, ia], [0, ib, ic, ia], [-ib, 0, -ic, ia]])
    np.testing.assert_allclose(Ainv, g["A_inv"], rtol=1e-12, atol=1e-6)
    mom, rot = orc.attitude_run(cfg, g["quat_xyzw"], g["omega"], g["cmd"])
    np.testing.assert_allclose(mom, g["moment"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(rot, g["rotor"], rtol=1e-10, atol=1e-7)
    assert (g["rotor"][20:30] == 0).any(), "fixture exercises the clamp at zero"
    mom32, rot32 = orc.attitude_run(DqlConfig(dtype=F32), g["quat_xyzw"], g["omega"], g["cmd"], dtype=0)
    # float32 tick: the yaw frame is normalised by THREE Newton steps from a second-order start (round 4; round 3's two left 2e-5 at a tilt of 45 deg
    # and 0.5 % at the fixture's most tilted sample, 60 deg): one tolerance over the whole fixture again
    x, y, z, w = g["quat_xyzw"].T
    tilt = np.degrees(np.arccos(np.sqrt(np.clip((1 - 2 * (y * y + z * z)) ** 2 + (2 * (x * y + w * z)) ** 2, 0, 1))))
    assert (tilt <= 45).sum() > 150 and tilt.max() > 55
    np.testing.assert_allclose(mom32, g["moment"], rtol=2e-4, atol=2e-6)
    # ... and the rotor speeds it commands (round 5: rot32 was computed and never asserted).  Compared as w^2 against the size S of the terms the
    # inverse allocation adds up (tests/fixture_checks.py check_g9_rotor_speeds has the derivation and the float64 / HIP legs)
    from fixture_checks import _alloc_scale
    S = _alloc_scale(DqlConfig(), g["cmd"][:, 3], g["moment"])
    err = np.abs(rot32 ** 2 - g["rotor"] ** 2).max(axis=1) / S
    assert err[tilt <= 55].max() <= 1e-6 and err.max() <= 5e-5


def test_g11_platform(golden_dir):
    g = np.load(golden_dir / "g11_platform.npz")
    out = orc.platform_run(DqlConfig(dtype=F64), 3000)
    np.testing.assert_allclose(out, g["rpm_launch"][:, 1:], rtol=0, atol=2e-11)
    out = orc.platform_run(DqlConfig(dtype=F64, mp_t_x=1.0), 3000)
    np.testing.assert_allclose(out, g["rpm_default"][:, 1:], rtol=0, atol=2e-11)
    out = orc.platform_run(DqlConfig(dtype=F64, trajectory=TRAJ_EIGHT), 3000)
    np.testing.assert_allclose(out, g["eight"][:, 1:], rtol=0, atol=2e-11)


def test_det_math_accuracy():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-7, 7, 4000), [0.0, 1e-9, np.pi / 2, np.pi, 2 * np.pi - 1e-12]])
    y = rng.uniform(-3, 3, len(x))
    s, c, a, lg = orc.det_math(x, y, dtype=1)
    np.testing.assert_allclose(s, np.sin(x), rtol=0, atol=3e-16)
    np.testing.assert_allclose(c, np.cos(x), rtol=0, atol=3e-16)
    np.testing.assert_allclose(a, np.arctan2(y, x), rtol=0, atol=5e-16)
    m = np.abs(x) > 1e-30
    np.testing.assert_allclose(lg[m], np.log(np.abs(x[m])), rtol=4e-16, atol=5e-16)
    s, c, a, lg = orc.det_math(x, y, dtype=0)
    x32 = x.astype(np.float32).astype(np.float64); y32 = y.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(s, np.sin(x32), rtol=0, atol=2e-7)
    np.testing.assert_allclose(c, np.cos(x32), rtol=0, atol=2e-7)
    np.testing.assert_allclose(a, np.arctan2(y32, x32), rtol=0, atol=5e-7)
    m = np.abs(x32) > 1e-30
    np.testing.assert_allclose(lg[m], np.log(np.abs(x32[m])), rtol=2e-7, atol=1e-6)


def test_philox_known_answer():
    """Random123 known-answer vectors for philox4x32-10."""
    np.testing.assert_array_equal(orc.philox((0, 0, 0, 0), (0, 0)), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8])
    np.testing.assert_array_equal(orc.philox((0xffffffff,) * 4, (0xffffffff, 0xffffffff)),
                                  [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd])
    np.testing.assert_array_equal(orc.philox((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)),
                                  [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])


def test_g7_npy_layout(golden_dir):
    meta = json.loads((golden_dir / "g7_npy.json").read_text())
    for name in ("Q_table_a.npy", "Q_table_b.npy", "state_action_count.npy"):
        assert meta[name]["size"] == 22808
        hdr = bytes.fromhex(meta[name]["header_hex"])
        assert hdr[:6] == b"\x93NUMPY" and hdr[6:8] == b"\x01\x00"
        assert b"'descr': '<f8', 'fortran_order': False, 'shape': (5, 3, 3, 3, 7, 3), }" in hdr


# ---- G12: the 100 Hz manager tick (ManagerNode.publish_obs + ObservationUtils), pins a20 and quirk B19 ----
def _g12_cfg(noise_sd, quirks=None):
    from dql_multirotor_landing_amd.config import Q_REFERENCE
    return DqlConfig(dtype=F64, two_axis=1, noise_pos_sd=float(noise_sd[0]), noise_vel_sd=float(noise_sd[1]), quirks=Q_REFERENCE if quirks is None else quirks)


def test_g12_manager_tick_matches_reference(golden_dir):
    """What one manager tick publishes, against the reference's own ManagerNode.publish_obs over 300 scripted ticks: relative
    position / velocity = platform - drone in the yaw-only frame, acceleration through the Kalman filter from the UN-noised
    velocity, PID plant states (v_z = -rel_v_z, yaw of q_drone q_platform^-1), and the platform set-point published by the SAME
    tick (the observation still sees the previous one).  Tolerances: the rotation helpers on the reference side are stand-ins
    for tf (as in G9) and the frame rotation is a Newton 1/sqrt here, so 1e-12 instead of bit equality."""
    z = np.load(golden_dir / "g12_manager.npz")
    for tag in ("a_noise0", "c_yaw"):
        got = orc.manager_run(_g12_cfg(z[f"{tag}_noise_sd"]), z[f"{tag}_in"], z[f"{tag}_contact"])
        ref = z[f"{tag}_out"]
        np.testing.assert_allclose(got[:, :4], ref[:, :4], rtol=0, atol=2e-12, err_msg=tag + " p / v")
        np.testing.assert_allclose(got[:, 4:6], ref[:, 4:6], rtol=1e-9, atol=2e-10, err_msg=tag + " acceleration")   # (v - v0) / dt: cancellation
        np.testing.assert_allclose(got[:, 6], ref[:, 6], rtol=0, atol=0, err_msg=tag + " v_z plant state")
        np.testing.assert_allclose(got[:, 7], ref[:, 7], rtol=0, atol=1e-12, err_msg=tag + " yaw plant state")
        np.testing.assert_allclose(got[:, 8:], ref[:, 8:], rtol=0, atol=1e-11, err_msg=tag + " platform set-point")  # G11's tolerance
    # the published set-point is NOT what this tick's observation used: rel_p_x = (platform as reported) - drone
    a_in, a_out = z["a_noise0_in"], z["a_noise0_out"]
    np.testing.assert_allclose(a_out[:, 0], a_in[:, 10] - a_in[:, 0], rtol=0, atol=1e-15)
    assert np.abs(a_out[:, 8] - a_in[:, 10]).max() > 1e-5  # the fresh set-point differs (by the extrapolation error) and is not what was observed


def test_g12_quirk_b19_frozen_acceleration_reference(golden_dir):
    """B19, asserted from REFERENCE OUTPUT: ObservationUtils sets last_velocity / last_timestep on its first call and never again
    (pkg/observation_utils.py:137-150), so rel_a = (v_i - v_0) / (t_i - t_0) — the mean acceleration since the node started —
    not the finite difference of successive samples.  The oracle reproduces it with the quirk bit and departs without."""
    from dql_multirotor_landing_amd.config import Q_FROZEN_ACC_REFERENCE, Q_REFERENCE
    z = np.load(golden_dir / "g12_manager.npz")
    rel, out = z["a_noise0_rel"], z["a_noise0_out"]
    i = np.arange(1, len(rel))
    frozen = (rel[1:, 3] - rel[0, 3]) / (0.01 * i)
    np.testing.assert_allclose(out[1:, 4], frozen, rtol=1e-15, atol=0)   # R = 0: the filter passes it through as x + 1 * (z - x), one rounding
    successive = np.diff(rel[:, 3]) / 0.01
    assert np.abs(out[1:, 4] - successive)[50:].mean() > 0.05            # and clearly not a per-tick finite difference
    assert out[0, 4] == 0.0
    paper = orc.manager_run(_g12_cfg(z["a_noise0_noise_sd"], quirks=Q_REFERENCE & ~Q_FROZEN_ACC_REFERENCE), z["a_noise0_in"], z["a_noise0_contact"])
    np.testing.assert_allclose(paper[1:, 4], successive, rtol=1e-9, atol=1e-9)


def test_g12_noise_feeds_the_message_not_the_filter(golden_dir):
    """With observation noise (0.25 m, 0.1 m/s: the code defaults, scripts/manager_node.py:83-88) the reference adds N(0, sd) to the
    published position and velocity only; the acceleration estimate runs on the clean velocity with R = sd_v^2
    (pkg/filters.py:49) — so rel_a is deterministic and must match to rounding, while the noise itself is compared in distribution."""
    z = np.load(golden_dir / "g12_manager.npz")
    sd = z["b_noise_noise_sd"]
    got = orc.manager_run(_g12_cfg(sd), z["b_noise_in"], z["b_noise_contact"], seed=5)
    np.testing.assert_allclose(got[:, 4:6], z["b_noise_out"][:, 4:6], rtol=1e-9, atol=1e-10)
    ref_noise = z["b_noise_noise"]
    assert abs(ref_noise[:, 0].std() / sd[0] - 1) < 0.15 and abs(ref_noise[:, 3].std() / sd[1] - 1) < 0.15
    mine_p = (got[:, 0] - z["b_noise_rel"][:, 0]) / sd[0]; mine_v = (got[:, 2] - z["b_noise_rel"][:, 3]) / sd[1]
    for m in (mine_p, mine_v):
        assert abs(m.std() - 1) < 0.15 and abs(m.mean()) < 0.2
    assert abs(np.corrcoef(mine_p, mine_v)[0, 1]) < 0.2


# ---- G13: TrainingLandingEnv / SimulationLandingEnv reset() and step() against a fake Gazebo playing back a recorded flight ----
G13_CASES = {"train0": dict(working_curriculum_step=0, t_max=4.0), "train2": dict(working_curriculum_step=2, t_max=4.0),
             "sim4": dict(working_curriculum_step=4, t_max=6.0, vz_setpoint=-0.4, init_uniform=2, goal_logic=0, z_init=4.0)}
G13_SEED = {"train0": 1300, "train2": 1302, "sim4": 1304}


def _g13_fly(engine_cls, tag, z):
    """this build's simulator (oracle or HIP engine, float64, one env) on the fixture's seed and actions -> per-period signals + outputs"""
    o = engine_cls(DqlConfig(dtype=F64, **G13_CASES[tag]), 1, seed=G13_SEED[tag])
    names, inames = o.field_names(), o.field_names(True)
    sig, res = [], []
    for a in z[f"{tag}_actions"]:
        o.step(np.array([a], dtype=np.uint8))
        reals, ints = o.get_fields()
        g = lambda k: float(reals[names.index(k)][0]); gi = lambda k: int(ints[inames.index(k)][0])
        sig.append([g("obs_p_x"), g("obs_p_y"), g("obs_v_x"), g("obs_v_y"), g("obs_a_x"), g("obs_a_y"), g("qw"), g("qx"), g("qy"), g("qz"), g("pz"),
                    float(bool(gi("flags") & 16))])
        res.append([float(bool(gi("flags") & 8)), float(bool(gi("flags") & 1)), gi("idx_x"), gi("code"), g("reward"), g("pitch_sp")])
    return np.array(sig), np.array(res)


def _g13_check(tag, z, sig, res):
    # (1) the played-back flight IS this simulator's flight (if this fails after a deliberate arithmetic change: regenerate the fixture)
    rec = np.column_stack([z[f"{tag}_rec_obs"], z[f"{tag}_rec_quat"], z[f"{tag}_rec_z"], z[f"{tag}_rec_contact"]])
    np.testing.assert_array_equal(sig, rec, err_msg="simulator signals differ from the recorded flight: regenerate tests/golden (make_golden.py)")
    np.testing.assert_array_equal(res, z[f"{tag}_sim"])
    # (2) what the reference's env.reset() / env.step() returned for those signals == what the fused step produced
    rows = z[f"{tag}_rows"]
    pack = lambda t: int((((t[0] * 3 + t[1]) * 3 + t[2]) * 3 + t[3]) * 7 + t[4])
    assert len(rows) == len(res)
    for r, s in zip(rows, res):
        assert pack(r[2:7]) == int(s[2])                       # discrete state
        assert bool(r[0] == 0) == bool(s[0])                   # reset periods line up
        if r[0] == 1:
            assert bool(r[13]) == bool(s[1])                   # done <=> "Termination condition" in info
            if tag != "sim4":
                assert r[12] == s[4]                           # reward, float64 ==
                assert int(r[14]) == int(s[3])                 # CheckResult
    assert rows[:, 13].sum() >= 3, "the script must contain several terminated episodes"


def test_g13_env_reset_step_sequencing_and_outputs(golden_dir):
    """a17-a19: the reference's env classes, driven by a fake Gazebo that plays back a flight of THIS simulator, return per agent
    period exactly what the fused step computes (state index, reward, done, CheckResult), reset periods included; their service /
    topic call order is the one the fused kernel's period follows (set-point first, then one agent period of simulation, then
    the fresh pose + the latest latched observation)."""
    import json
    z = np.load(golden_dir / "g13_env.npz")
    for tag in G13_CASES:
        sig, res = _g13_fly(orc.Oracle, tag, z)
        _g13_check(tag, z, sig, res)
    calls = json.loads((golden_dir / "g13_env_calls.json").read_text())
    for tag in G13_CASES:
        assert calls[tag]["reset_calls"] == ["srv:pause_physics", "srv:get_model_state:moving_platform", "srv:set_model_state", "pub:reset_simulation",
                                             "srv:unpause_physics", "sleep", "srv:pause_physics", "srv:get_model_state:hummingbird"]
        assert calls[tag]["step_calls"] == ["pub:action_to_interface", "srv:unpause_physics", "sleep", "srv:pause_physics", "srv:get_model_state:hummingbird"]


def test_g13_reset_placement_arithmetic(golden_dir):
    """TrainingLandingEnv.reset places the drone at clip(x0 + mp_x, mp_x +- p_max) (x0 ~ N(0, p_max / 3) at level 0, U(-p_max, p_max)
    above), y = 0, z = z_init; SimulationLandingEnv.reset at clip(mp_x - x0, +-p_max) with its y offset multiplied by 0 (B16).
    Same arithmetic in the simulator's reset (init_uniform 0 / 2), bit for bit."""
    z = np.load(golden_dir / "g13_env.npz")
    for tag, mode in (("train0", 0), ("train2", 0), ("sim4", 2)):
        pl = z[f"{tag}_placements"]   # x0, y0, platform x, placed x, y, z
        got = orc.place(DqlConfig(dtype=F64, init_uniform=mode), pl[:, 0], pl[:, 2])
        np.testing.assert_array_equal(got, pl[:, 3])
        assert (pl[:, 4] == 0).all() and (pl[:, 5] == 4.0).all()
    pl = z["sim4_placements"]
    assert (np.abs(pl[:, 3]) <= 4.5).all() and np.abs(pl[:, 3] - (pl[:, 2] - pl[:, 0])).min() == 0.0
    # the absolute clip is reached by construction in some draw of a longer series
    x0 = np.linspace(-4.5, 4.5, 19); mp = np.full(19, 1.9)
    got = orc.place(DqlConfig(dtype=F64, init_uniform=2), x0, mp)
    np.testing.assert_array_equal(got, np.clip(mp - x0, -4.5, 4.5))
    assert (got == 4.5).any()


def test_g13_simulation_env_y_state(golden_dir):
    """SimulationLandingEnv returns a second tuple for the y axis, discretised from rel_*_y and ROLL (pkg/mdp.py:625-782); the
    reference never flies y (B16), so it is the state of a drone sitting at y = 0: reproduced from the recorded signals."""
    z = np.load(golden_dir / "g13_env.npz")
    rows, obs, q = z["sim4_rows"], z["sim4_rec_obs"], z["sim4_rec_quat"]
    roll = np.arctan2(2 * (q[:, 2] * q[:, 3] + q[:, 0] * q[:, 1]), 1 - 2 * (q[:, 1] ** 2 + q[:, 2] ** 2))
    idx = orc.discretise(DqlConfig(dtype=F64, working_curriculum_step=4), obs[:, 1], obs[:, 3], obs[:, 5], roll)
    sy = rows[:, 7:12].astype(int)
    np.testing.assert_array_equal(idx, (((sy[:, 0] * 3 + sy[:, 1]) * 3 + sy[:, 2]) * 3 + sy[:, 3]) * 7 + sy[:, 4])


def test_lazy_noise_equals_eager_noise():
    """Round 3 shortens the fused step by drawing the observation noise only at the LAST manager tick of an agent period (the only
    draw the MDP ever reads: the noise sits on the latched p / v, which every manager tick overwrites).  Pinned here against drawing
    at every tick, in both dtypes, with periods of 21 and 22 ticks and 4 or 5 manager ticks: every field, the tables and the counters."""
    from dql_multirotor_landing_amd.config import DqlConfig, F32, F64
    from oracle.oracle import Oracle
    for dtype in (F32, F64):
        kw = dict(dtype=dtype, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1, t_max=4.0)
        lazy, eager = Oracle(DqlConfig(**kw), 96, seed=5), Oracle(DqlConfig(**kw), 96, seed=5)
        try:
            eager.set_option("eager_noise", 1)
            eager.train_steps(60, 0.6)
            eager.set_option("eager_noise", 0)
            lazy.train_steps(60, 0.6)
        finally:
            eager.set_option("eager_noise", 0)
        for a, b in zip(lazy.get_fields(), eager.get_fields()):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(lazy.qa, eager.qa)
        np.testing.assert_array_equal(lazy.count, eager.count)
        assert lazy.stats_dict() == eager.stats_dict() and lazy.stats_dict()["episodes"] > 0
        reals, _ = lazy.get_fields()
        names = lazy.field_names()
        assert np.abs(reals[names.index("obs_p_x")] - (reals[names.index("mp_x")] - reals[names.index("px")])).max() > 0.05  # the noise is there
