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
    np.testing.assert_allclose(mom32, g["moment"], rtol=2e-4, atol=2e-6)


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
