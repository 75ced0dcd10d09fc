"""Stateless C-ABI operators on the GPU against the golden vectors of the reference (bit-exact in float64)."""
import numpy as np
import pytest

from dql_multirotor_landing_amd.config import DqlConfig, F64, N_CELLS

pytestmark = pytest.mark.gpu


def unpack(idx):
    idx = np.asarray(idx)
    return np.stack([idx // 189, (idx // 63) % 3, (idx // 21) % 3, (idx // 7) % 3, idx % 7], axis=-1)


def pack(s):
    s = np.asarray(s).astype(np.int64)
    return ((((s[..., 0] * 3 + s[..., 1]) * 3 + s[..., 2]) * 3 + s[..., 3]) * 7 + s[..., 4]).astype(np.int32)


@pytest.fixture(scope="module")
def ops():
    from dql_multirotor_landing_amd import ops
    return ops


@pytest.mark.parametrize("level", range(5))
def test_discretise_golden(ops, golden_dir, level):
    g = np.load(golden_dir / "g1_discretise.npz")
    x = g[f"in_{level}"]
    idx = ops.discretise(DqlConfig(working_curriculum_step=level, dtype=F64), x[:, 0], x[:, 1], x[:, 2], x[:, 3])
    np.testing.assert_array_equal(unpack(idx), g[f"state_{level}"])


@pytest.mark.parametrize("level", [0, 3])
def test_mdp_trace_golden(ops, golden_dir, level):
    t = np.load(golden_dir / "g2_traces.npz")[f"trace_{level}"]
    cfg = DqlConfig(working_curriculum_step=level, dtype=F64)
    ms = np.zeros((8, 1)); ms[7] = 8
    prev = np.array([-1], dtype=np.int32)
    for row in t[:700]:
        op, act = int(row[0]), int(row[1])
        obs = np.array([row[2], row[3], row[4], row[5], row[6], row[7], row[8]]).reshape(7, 1)
        if op == 0:
            ms[0] = 0.0; ms[4] = 0.0; ms[5] = 0; ms[6] = 0; ms[7] = 8
            prev = ops.discretise(cfg, obs[0], obs[2], obs[3], obs[4]).astype(np.int32)
            continue
        ms, idx, rew, done = ops.mdp_transition(cfg, [act], obs, ms, prev)
        np.testing.assert_array_equal(unpack(idx)[0], row[9:14].astype(int))
        assert int(ms[7, 0]) == int(row[14]) and rew[0] == row[15] and int(done[0]) == int(row[16])
        assert ms[0, 0] == row[17] and ms[4, 0] == row[18]
        prev = idx.astype(np.int32)


def test_agent_golden(ops, golden_dir):
    g = np.load(golden_dir / "g4_agent.npz")
    qa = np.zeros(N_CELLS); qb = np.zeros(N_CELLS); cnt = np.zeros(N_CELLS)
    sa = g["upd_sa"]
    ops.agent_update(qa, qb, cnt, pack(sa[:, :5]) * 3 + sa[:, 5], pack(g["upd_ns"]), g["upd_alpha"], 0.99, g["upd_reward"])
    np.testing.assert_array_equal(qa, g["upd_Qa"].ravel())
    np.testing.assert_array_equal(cnt, g["upd_count"].ravel())
    assert not qb.any()
    A = np.load(golden_dir / "assets" / "Q_table_a.npy"); B = np.load(golden_dir / "assets" / "Q_table_b.npy")
    np.testing.assert_array_equal(ops.agent_predict(A, B, np.arange(945)), g["predict_actions"])


# ---- G12: the manager tick of the fused kernel (manager_states + manager_obs) through dql_manager_run ----
def test_manager_run_golden_and_oracle(ops, golden_dir):
    """a20 on the GPU: the device functions of the fused step's 100 Hz manager tick, replayed over the reference's scripted series:
    == the oracle bit for bit (float64 and float32), and == ManagerNode.publish_obs / ObservationUtils within the tolerances of
    tests/test_oracle_golden.py (rotation helpers of tf are stand-ins there); quirk B19 reproduced from reference output."""
    from dql_multirotor_landing_amd.config import F32, Q_FROZEN_ACC_REFERENCE, Q_REFERENCE
    from oracle import oracle as orc
    z = np.load(golden_dir / "g12_manager.npz")
    for tag in ("a_noise0", "b_noise", "c_yaw"):
        sd = z[f"{tag}_noise_sd"]
        for dtype in (F64, F32):
            cfg = DqlConfig(dtype=dtype, two_axis=1, noise_pos_sd=float(sd[0]), noise_vel_sd=float(sd[1]))
            got = ops.manager_run(cfg, z[f"{tag}_in"][None], z[f"{tag}_contact"][None], seed=5)[0]
            # the series index keys the noise stream: series 0 here == env id 0 in the oracle's replay
            want = orc.manager_run(cfg, z[f"{tag}_in"], z[f"{tag}_contact"], seed=5)
            np.testing.assert_array_equal(got, want, err_msg=f"{tag} dtype {dtype}: HIP != oracle")
        ref = z[f"{tag}_out"]
        got = ops.manager_run(DqlConfig(dtype=F64, two_axis=1, noise_pos_sd=float(sd[0]), noise_vel_sd=float(sd[1])), z[f"{tag}_in"][None], z[f"{tag}_contact"][None])[0]
        np.testing.assert_allclose(got[:, 4:6], ref[:, 4:6], rtol=1e-9, atol=2e-10)      # acceleration: clean velocity, Kalman R = sd^2, B19
        np.testing.assert_allclose(got[:, 6], ref[:, 6], rtol=0, atol=0)                  # v_z plant state
        np.testing.assert_allclose(got[:, 7], ref[:, 7], rtol=0, atol=1e-12)              # yaw plant state
        np.testing.assert_allclose(got[:, 8:], ref[:, 8:], rtol=0, atol=1e-11)            # platform set-point published by the tick
        if sd[0] == 0:
            np.testing.assert_allclose(got[:, :4], ref[:, :4], rtol=0, atol=2e-12)
    rel = z["a_noise0_rel"]
    i = np.arange(1, len(rel))
    got = ops.manager_run(DqlConfig(dtype=F64, two_axis=1), z["a_noise0_in"][None], z["a_noise0_contact"][None])[0]
    np.testing.assert_allclose(got[1:, 4], (rel[1:, 3] - rel[0, 3]) / (0.01 * i), rtol=1e-9, atol=1e-10)   # B19: frozen reference sample
    paper = ops.manager_run(DqlConfig(dtype=F64, two_axis=1, quirks=Q_REFERENCE & ~Q_FROZEN_ACC_REFERENCE), z["a_noise0_in"][None], z["a_noise0_contact"][None])[0]
    np.testing.assert_allclose(paper[1:, 4], np.diff(rel[:, 3]) / 0.01, rtol=1e-9, atol=1e-9)
    # several series at once: one lane each, own noise stream each
    many = ops.manager_run(DqlConfig(dtype=F64, two_axis=1, noise_pos_sd=0.25, noise_vel_sd=0.1), np.repeat(z["b_noise_in"][None], 70, axis=0),
                           np.repeat(z["b_noise_contact"][None], 70, axis=0), seed=9)
    assert many.shape == (70, 300, 12) and np.array_equal(many[:, :, 4:6], np.repeat(many[:1, :, 4:6], 70, axis=0)) and not np.array_equal(many[0, :, 0], many[1, :, 0])


def test_plant_run_equals_oracle(ops):
    """a26 / a27: the open-loop plant operator == its oracle twin bit for bit on random series (float64 and float32): tumbling attitudes,
    rotor commands above the speed limit, platforms under the vehicle.  (What pins BOTH against the reference's formulas:
    tests/test_plant_closed_forms.py.)"""
    from dql_multirotor_landing_amd.config import F32
    from oracle import oracle as orc
    rng = np.random.default_rng(11)
    ns, nt = 130, 120
    q = rng.normal(size=(ns, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    init = np.c_[rng.uniform(-3, 3, (ns, 2)), rng.uniform(0.4, 4, (ns, 1)), rng.uniform(-2, 2, (ns, 3)), q, rng.uniform(-3, 3, (ns, 3)),
                 rng.uniform(0, 838, (ns, 4)), rng.uniform(-3, 3, (ns, 2)), rng.uniform(-1.6, 1.6, (ns, 2))]
    init[:20, 0:2] = init[:20, 17:19] + rng.uniform(-0.3, 0.3, (20, 2)); init[:20, 2] = 0.5; init[:20, 3:6] = 0.0  # these start on the deck
    cmd = rng.uniform(0, 1000, (ns, nt, 4))
    for dtype in (F64, F32):
        cfg = DqlConfig(dtype=dtype)
        got = ops.plant_run(cfg, init, cmd)
        want = orc.plant_run(cfg, init, cmd)
        np.testing.assert_array_equal(got, want, err_msg=f"dtype {dtype}: HIP != oracle")
        assert got[:, :, 19].max() == 1.0 and got[:, :, 19].min() == 0.0  # some series touch the platform, some never do
    with pytest.raises(ValueError):
        ops.plant_run(DqlConfig(), init, -cmd)
    assert ops.plant_run(DqlConfig(), init[:0], cmd[:0]).shape == (0, nt, 20)


def test_place_golden(ops, golden_dir):
    """a17 / a19 placement arithmetic of reset() on the GPU == what the reference's reset() handed to /gazebo/set_model_state."""
    z = np.load(golden_dir / "g13_env.npz")
    for tag, mode in (("train0", 0), ("train2", 0), ("sim4", 2)):
        pl = z[f"{tag}_placements"]
        np.testing.assert_array_equal(ops.place(DqlConfig(dtype=F64, init_uniform=mode), pl[:, 0], pl[:, 2]), pl[:, 3])
    with pytest.raises(ValueError):
        ops.place(DqlConfig(init_uniform=3), [0.0], [0.0])


def test_resident_agent_equals_stateless_operators(ops):
    """dql_agent_* (tables resident on the device, arguments and results through pinned memory) == dql_agent_predict / dql_agent_update
    (tables shipped per call) == the oracle, over a random sequence in both update rules; the reported (new cell value, new counter, greedy
    action of the next state) are what the tables then hold."""
    import ctypes as C
    from dql_multirotor_landing_amd import _lib
    from dql_multirotor_landing_amd.config import Q_PAPER, Q_REFERENCE
    from oracle import oracle as orc
    lib = _lib.load()
    rng = np.random.default_rng(3)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for quirks in (Q_REFERENCE, Q_PAPER):
        qa, qb, cnt = rng.normal(size=2835), rng.normal(size=2835), rng.integers(0, 50, 2835).astype(np.float64)
        h = C.c_void_p()
        _lib.check(lib.dql_agent_create(0, C.byref(h)))
        _lib.check(lib.dql_agent_set_tables(h, p(qa), p(qb), p(cnt)))
        ra, rb, rc = qa.copy(), qb.copy(), cnt.copy()       # stateless replay
        oa, ob, oc = qa.copy(), qb.copy(), cnt.copy()       # oracle replay
        for _ in range(40):
            n = int(rng.integers(1, 6))
            sa = rng.integers(0, 2835, n).astype(np.int32); ns = rng.integers(0, 945, n).astype(np.int32)
            alpha, reward = rng.uniform(0.02, 1.0, n), rng.normal(size=n) * 5
            coin, done = rng.integers(0, 2, n).astype(np.uint8), rng.integers(0, 2, n).astype(np.uint8)
            q_new, c_new, nxt = np.zeros(n), np.zeros(n), np.zeros(1, np.uint8)
            _lib.check(lib.dql_agent_update_resident(h, p(sa), p(ns), p(alpha), 0.99, p(reward), n, quirks, p(coin), p(done), p(q_new), p(c_new), p(nxt)))
            ops.agent_update(ra, rb, rc, sa, ns, alpha, 0.99, reward, quirks=quirks, coin=coin, done=done)
            orc.agent_update(oa, ob, oc, sa, ns, alpha, 0.99, reward, quirks=quirks, coin=coin, done=done)
            assert c_new[-1] == rc[sa[-1]] and q_new[-1] in (ra[sa[-1]], rb[sa[-1]])
            assert nxt[0] == ops.agent_predict(ra, rb, ns[-1:])[0]
            idx = rng.integers(0, 945, 33).astype(np.int32); act = np.zeros(33, np.uint8)
            _lib.check(lib.dql_agent_predict_resident(h, p(idx), 33, p(act)))
            np.testing.assert_array_equal(act, ops.agent_predict(ra, rb, idx))
        ga, gb, gc = np.zeros(2835), np.zeros(2835), np.zeros(2835)
        _lib.check(lib.dql_agent_get_tables(h, p(ga), p(gb), p(gc)))
        for got, want, o in ((ga, ra, oa), (gb, rb, ob), (gc, rc, oc)):
            np.testing.assert_array_equal(got, want); np.testing.assert_array_equal(got, o)
        assert lib.dql_agent_predict_resident(h, p(np.array([945], np.int32)), 1, p(np.zeros(1, np.uint8))) == _lib.EINVAL
        _lib.check(lib.dql_agent_destroy(h))


@pytest.mark.parametrize("n", [200, 300, 20000])
def test_step_outputs_and_kernel_side_action_check(n):
    """dql_step_outputs == what the field getters say (state, reward, done, code, step count, cumulative reward, reset flag) in one round
    trip; an out-of-range action handed straight to the C ABI is refused before anything moves (up to 16 384 envs) or flown as "hold" and reported
    ONCE by the next outputs / stats call (beyond: no host loop over the actions).  The three sizes take the three paths of the pair dql_step / dql_step_outputs: 200 — one
    workgroup: actions read from pinned memory by the step kernel, results picked up when the kernel posts its sequence number; 300 —
    several workgroups: the stream is waited for; 20 000 (> DQL_ZERO_COPY_ENVS) — actions copied to the device first."""
    import ctypes as C
    from dql_multirotor_landing_amd import _lib
    from dql_multirotor_landing_amd.engine import Engine
    eng = Engine(DqlConfig(dtype=F64, t_max=2.0), n, seed=4)
    twin = Engine(DqlConfig(dtype=F64, t_max=2.0), n, seed=4)
    rng = np.random.default_rng(1)
    for k in range(70):
        a = rng.integers(0, 3, n).astype(np.uint8)
        eng.step(a); twin.step(a)
        o = eng.step_outputs()
        reals, ints = eng.get_fields()
        nm, im = eng.field_names(), eng.field_names(True)
        np.testing.assert_array_equal(o["idx_x"], ints[im.index("idx_x")])
        np.testing.assert_array_equal(o["reward"], reals[nm.index("reward")])
        np.testing.assert_array_equal(o["cumulative_reward"], reals[nm.index("cum_x")])
        np.testing.assert_array_equal(o["done"], ints[im.index("flags")] & 1)
        np.testing.assert_array_equal(o["was_reset"], (ints[im.index("flags")] >> 3) & 1)
        np.testing.assert_array_equal(o["code"], ints[im.index("code")])
        np.testing.assert_array_equal(o["step_count"], ints[im.index("step_count")])
    assert o["done"].sum() + o["was_reset"].sum() > 0
    bad = np.full(n, 2, dtype=np.uint8); bad[7] = 3; bad[9] = 1 << 2      # ax = 3; a y action in an x-axis config
    if n <= 16384:
        # small batches (the single-env drop-in path among them) are checked while the actions are staged: refused BEFORE anything is flown
        before = eng.get_fields()
        assert eng.lib.dql_step(eng._h, bad.ctypes.data_as(C.c_void_p)) == _lib.EINVAL
        after = eng.get_fields()
        np.testing.assert_array_equal(before[0], after[0]); np.testing.assert_array_equal(before[1], after[1])
        only_y = np.full(n, 2, dtype=np.uint8); only_y[9] = 1 << 2
        assert eng.lib.dql_step(eng._h, only_y.ctypes.data_as(C.c_void_p)) == _lib.EINVAL   # "Cannot move in the y direction while training"
        eng.step_outputs()                                                # nothing pending: no error is reported later either
    else:
        assert eng.lib.dql_step(eng._h, bad.ctypes.data_as(C.c_void_p)) == 0  # accepted: nobody loops over 20 000 envs on the host
        with pytest.raises(ValueError, match="2 action"):
            eng.step_outputs()
        eng.step_outputs()                                                   # reported once
        hold = np.full(n, 2, dtype=np.uint8); hold[9] = 0
        twin.step(hold)                                                      # ax = 3 flew as "hold"; env 9's x action (0) was flown, its y action had nothing to act on
        np.testing.assert_array_equal(eng.get_fields()[0], twin.get_fields()[0])
    with pytest.raises(ValueError, match="y direction"):
        eng.step(np.full(n, 2 | 1 << 2, dtype=np.uint8))                 # Engine.step: a y action in an x-axis config, as the reference raises
    with pytest.raises(ValueError):
        eng.step(np.full(n, 3, dtype=np.uint8))                          # the Python layer still refuses up front, as the reference's step() raises
    eng.close(); twin.close()


def test_float32_tick_square_root_is_correctly_rounded_on_its_whole_domain(ops):
    """The float32 tick takes the rotor commands' square roots as v_rsq_f32 + one residual correction (1 transcendental + 4 full-rate instructions; rounds 3 - 5
    ran a Goldschmidt step in between: 1 + 7, tools/micro/sqrt_variants.hip) and the oracle as sqrtf(): bit-for-bit parity needs the former to be THE correctly rounded root.  Exhaustive:
    every float32 from 1e-30 (the tick clamps there) to FLT_MAX — 2.1e9 inputs; and the reason for the clamp: below 2^-103 it is not."""
    assert ops.selftest_sqrt(1e-30, 3.4028234663852886e38) == 0
    assert ops.selftest_sqrt(2.0 ** -102, 1e-30) == 0
    assert ops.selftest_sqrt(2.0 ** -126, 2.0 ** -104) > 0


def test_agent_mirror_argument_checks():
    """dql_agent_mirror_* refuse what would read or write outside the caller's tables (include/dql.h): null arrays, level counts outside
    1..5, cells / states beyond the n_levels levels the arrays hold; a refused call changes nothing."""
    import ctypes as C
    from dql_multirotor_landing_amd import _lib
    from dql_multirotor_landing_amd.config import Q_REFERENCE
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.dql_agent_create(0, C.byref(h)))
    n = 3
    qa, qb, cnt = np.zeros(n * 567), np.ones(n * 567), np.zeros(n * 567)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    act = C.c_uint8(9)
    upd = lambda levels, sa, ns: lib.dql_agent_mirror_update(h, p(qa), p(qb), p(cnt), levels, sa, ns, 0.5, 0.99, 1.0, Q_REFERENCE, 0, 0)
    assert lib.dql_agent_mirror_predict(h, None, p(qb), p(cnt), n, 0, C.byref(act)) == _lib.EINVAL
    assert lib.dql_agent_mirror_predict(h, p(qa), p(qb), p(cnt), 0, 0, C.byref(act)) == _lib.EINVAL
    assert lib.dql_agent_mirror_predict(h, p(qa), p(qb), p(cnt), 6, 0, C.byref(act)) == _lib.EINVAL
    assert lib.dql_agent_mirror_predict(h, p(qa), p(qb), p(cnt), n, n * 189, C.byref(act)) == _lib.EINVAL   # first state of level 3: not in a 3-level table
    assert lib.dql_agent_mirror_predict(h, p(qa), p(qb), p(cnt), n, -1, C.byref(act)) == _lib.EINVAL
    assert lib.dql_agent_mirror_predict(h, p(qa), p(qb), p(cnt), n, 0, None) == _lib.EINVAL
    assert upd(n, n * 567, 0) == _lib.EINVAL and upd(n, 0, n * 189) == _lib.EINVAL and upd(n, -1, 0) == _lib.EINVAL and upd(7, 0, 0) == _lib.EINVAL
    assert act.value == 9 and not qa.any() and not cnt.any()
    _lib.check(upd(n, 5, 7)); _lib.check(lib.dql_agent_mirror_predict(h, p(qa), p(qb), p(cnt), n, 7, C.byref(act)))
    assert qa[5] == 0.5 * (1.0 + 0.99 * 0.0 * 0 - 0.0) and cnt[5] == 1.0 and act.value in (0, 1, 2)
    _lib.check(lib.dql_agent_destroy(h))


# ---- round 5: every control-side function of the tick through the C ABI against the reference's own outputs, in BOTH dtypes ----
# (the float32 legs are what ties the arithmetic every throughput figure runs on to reference-held data; bounds: tests/fixture_checks.py)
@pytest.fixture(scope="module")
def hip_be():
    import fixture_checks as fc
    return fc, fc.HipBackend(), fc.OracleBackend()


@pytest.mark.parametrize("dtype", [F64, 0])
def test_g8_filters_and_pid_on_hip(hip_be, dtype):
    """pkg/filters.py:19-109 + pkg/pid.py:62-104 (G8): dql_butterworth_run / dql_kalman_run / dql_pid_run"""
    fc, hip, _ = hip_be
    fc.check_g8_butterworth(hip, dtype)
    fc.check_g8_kalman(hip, dtype)
    fc.check_g8_pid(hip, dtype)


@pytest.mark.parametrize("dtype", [F64, 0])
def test_g9_rotor_speeds_on_hip(hip_be, dtype):
    """pkg/attitude_controller.py:107-156 (G9) down to the rotor command, v_rsq + residual-correction root and med3 clamp included: dql_attitude_run"""
    fc, hip, _ = hip_be
    fc.check_g9_rotor_speeds(hip, dtype)
    if dtype == 0:
        fc.check_g9_xonly_form(hip)


@pytest.mark.parametrize("dtype,carry", [(F64, 0), (0, 0), (0, 4), (0, 5)])
def test_g11_platform_on_hip(hip_be, dtype, carry):
    """pkg/moving_platform.py:87-127 (G11), incl. the rotation-carried sine / cosine of the float32 step: dql_platform_run"""
    fc, hip, _ = hip_be
    fc.check_g11_platform(hip, dtype, carry)


def test_g12_manager_tick_f32_on_hip(hip_be):
    """scripts/manager_node.py:192-214 + pkg/observation_utils.py:99-158 (G12) in float32 — Kalman fixed-point shortcut (the covariance
    reaches its fixed point ~100 ticks into the 300-tick series) and noise draws included: dql_manager_run"""
    fc, hip, _ = hip_be
    fc.check_g12_manager_f32(hip)


@pytest.mark.parametrize("dtype", [F64, 0])
def test_tick_operators_equal_the_oracle_bit_for_bit(hip_be, golden_dir, dtype):
    """same dtype, same operation sequence: HIP == oracle exactly, for every replay operator (what makes the oracle's fixture pins the kernel's)"""
    fc, hip, orc = hip_be
    g8 = np.load(golden_dir / "g8_filters.npz"); g9 = np.load(golden_dir / "g9_attitude.npz")
    cfg = DqlConfig(dtype=dtype)
    np.testing.assert_array_equal(hip.butterworth_run(cfg, g8["bw_in"]), orc.butterworth_run(cfg, g8["bw_in"]))
    flags = np.array([(i % 17 == 0) for i in range(120)], dtype=np.uint8)
    ck = DqlConfig(dtype=dtype, noise_vel_sd=0.1)
    np.testing.assert_array_equal(hip.kalman_run(ck, g8["kf_vel_r01"], flags), orc.kalman_run(ck, g8["kf_vel_r01"], flags))
    for tag in ("vz", "yaw"):
        a, b = hip.pid_run(cfg, g8[f"pid_{tag}_params"], g8[f"pid_{tag}_state"]), orc.pid_run(cfg, g8[f"pid_{tag}_params"], g8[f"pid_{tag}_state"])
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(hip.attitude_run(cfg, g9["quat_xyzw"], g9["omega"], g9["cmd"]), orc.attitude_run(cfg, g9["quat_xyzw"], g9["omega"], g9["cmd"]))
    for carry in ((0, 4, 5) if dtype == 0 else (0,)):
        np.testing.assert_array_equal(hip.platform_run(cfg, 1000, carry), orc.platform_run(cfg, 1000, carry))


def test_tick_operator_argument_checks(ops):
    with pytest.raises(ValueError):
        ops.pid_run(DqlConfig(), [2.0, 0.5, 0.3, -4, 4, 1, 0.2], np.zeros(10))       # Kd != 0: launch/drone.launch:37,51 has none
    with pytest.raises(ValueError):
        ops.attitude_run(DqlConfig(dtype=F64), np.tile([0, 0, 0, 1.0], (2, 1)), np.zeros((2, 3)), np.zeros((2, 4)), xonly=1)
    with pytest.raises(ValueError):
        ops.attitude_run(DqlConfig(dtype=0), np.tile([0, 0, 0, 1.0], (2, 1)), np.zeros((2, 3)), np.array([[0.1, 0, 0, 7.0]] * 2), xonly=1)
    with pytest.raises(ValueError):
        ops.platform_run(DqlConfig(dtype=F64), 10, carry=5)
    assert ops.platform_run(DqlConfig(), 0).shape == (0, 4)
